"""Debug probe: one-layer encoder error against the oracle with parts of the weights neutralised."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd import weights as W
from hybrid_rag_system_amd.engine import Engine
from oracle import encoder_oracle as EO
B, S = 8, 32
cfg = W.EncoderConfig(layers=1)
eng = Engine(0)
rng = np.random.default_rng(1)
ids = rng.integers(1000, 30000, (B, S)).astype(np.int32)
lens = np.full(B, S, np.int32)
def run(tag, edit):
    w = W.generate(cfg, 2026)
    edit(w)
    enc = eng.encoder(w, cfg)
    want = EO.forward(w, ids, lens, cfg).numpy()
    got = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
    enc.close()
    print(f"{tag:40s} max {np.abs(got-want).max():.5f} cos {(got*want).sum(1).min():.6f}")
def ident_ln(names):
    def f(w):
        for n in names:
            w[n + "_g"][:] = 1; w[n + "_b"][:] = 0
    return f
run("as generated", lambda w: None)
run("emb LN identity", ident_ln(["emb_ln"]))
run("ln2 identity", ident_ln(["l0.ln2"]))
run("emb+ln2 identity", ident_ln(["emb_ln", "l0.ln2"]))
run("all LN identity", ident_ln(["emb_ln", "l0.ln1", "l0.ln2"]))
def zero(names):
    def f(w):
        for n in names: w[n][:] = 0
    return f
run("no attention (o_w,o_b=0)", zero(["l0.o_w", "l0.o_b"]))
run("no ffn (d_w,d_b=0)", zero(["l0.d_w", "l0.d_b"]))
run("no attention, no ffn", zero(["l0.o_w", "l0.o_b", "l0.d_w", "l0.d_b"]))
