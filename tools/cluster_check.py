"""Exactness and fallback behaviour on a topically ordered (clustered) corpus."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd.engine import Engine
from oracle import search_oracle as so
eng = Engine(0)
rng = np.random.default_rng(5)
for csize, noise in ((500, 0.3), (500, 0.05), (3000, 0.2), (6000, 0.2)):
    d, ncl = 384, 120_000 // csize
    centers = rng.standard_normal((ncl, d)).astype(np.float32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    C = np.repeat(centers, csize, axis=0) + noise * rng.standard_normal((ncl * csize, d)).astype(np.float32) / np.sqrt(d)
    C = (C / np.linalg.norm(C, axis=1, keepdims=True)).astype(np.float32)
    Q = centers[rng.integers(0, ncl, 64)] + 0.1 * rng.standard_normal((64, d)).astype(np.float32) / np.sqrt(d)
    Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    want = so.topk_keys(Q, C, 10, "l2")
    for shadow in ("i8", "bf16"):
        c = eng.corpus(torch.from_numpy(C).cuda(), shadow=shadow)
        _, _, keys = eng.score_topk(c, torch.from_numpy(Q).cuda(), 10, mode="mfma", want_keys=True)
        torch.cuda.synchronize()
        ok = np.array_equal(keys.cpu().numpy().view(np.uint64), want)
        print(f"cluster {csize} noise {noise} {shadow}: exact={ok} {eng.score_stats()}", flush=True)
        c.close()
