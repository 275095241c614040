"""Encode fixed inputs and save the embeddings: python tools/enc_dump.py OUT.npy (compare two builds of the library)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd import weights as W
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0); cfg = W.EncoderConfig(); enc = eng.encoder(W.generate(cfg, 2026), cfg)
outs = []
for B, S, seed in ((16, 256, 1), (64, 256, 2), (40, 128, 3), (1024, 256, 4)):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, 30000, (B, S)).astype(np.int32)
    lens = rng.integers(3, S + 1, B).astype(np.int32); lens[0] = S
    for r in range(3):  # repeated calls must agree with each other too
        o = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()); torch.cuda.synchronize()
        outs.append(o.cpu().numpy())
np.save(sys.argv[1], np.concatenate(outs))
print("saved", sys.argv[1], sum(len(o) for o in outs))
