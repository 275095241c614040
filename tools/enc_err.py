"""Encoder error profile against the fp32 oracle: max |delta|, minimum cosine, and max |delta| per block of 32 features.
usage: python tools/enc_err.py [B S [layers]]   (HRS_LIBRARY selects another build of the library)"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd import weights as W
from hybrid_rag_system_amd.engine import Engine
from oracle import encoder_oracle as EO
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 256)
cfg = W.EncoderConfig()
if len(sys.argv) > 3:
    cfg = W.EncoderConfig(layers=int(sys.argv[3]))
w = W.generate(cfg, 2026)
eng = Engine(0); enc = eng.encoder(w, cfg)
rng = np.random.default_rng(B * 1000 + S)
ids = rng.integers(1000, 30000, (B, S)).astype(np.int32)
lens = rng.integers(3, S + 1, B).astype(np.int32); lens[0] = S
want = EO.forward(w, ids, lens, cfg).numpy()
got = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
d = np.abs(got - want)
print("layers", cfg.layers, "max", d.max(), "mean", d.mean(), "min cos", (got * want).sum(1).min())
print("per 32-feature block max:", np.round(d.reshape(B, 12, 32).max(axis=(0, 2)), 4))
print("per sequence max:", np.round(d.max(axis=1), 4), "lens", lens)
bad = np.where(d.max(axis=1) > 0.02)[0]
print("sequences with max error > 0.02:", bad.tolist()[:64], "of", B)
