"""Small-batch scoring calls, for a per-kernel trace: 1M x 384 cosine top-10 at batch 256 (sequential calls),
the same at 10M rows, and the exact scan at batch 4 over 10M rows.  usage: python tools/small_batch.py [reps]"""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
eng = Engine(0)
D, K = 384, 10


def corpus(N):
    m = torch.empty((N, D), dtype=torch.float32, device='cuda')
    for r0 in range(0, N, 1 << 20):
        nr = min(1 << 20, N - r0)
        eng.synth_rows(1235, r0, nr, D, out=m[r0:r0 + nr])
    return m, eng.corpus(m)


def timed(c, q, space, mode):
    eng.score_topk(c, q, K, space=space, mode=mode)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        eng.score_topk(c, q, K, space=space, mode=mode)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


for N in (1_000_000, 10_000_000):
    m, c = corpus(N)
    for B in (256,) if N == 1_000_000 else (256, 4):
        q = eng.synth_rows(4321, 0, B, D)
        for mode in (("mfma",) if B > 4 else ("exact", "mfma")):
            ms = timed(c, q, "cosine", mode)
            hbm = (N * D * (4 if mode == "exact" else 1)) / (ms * 1e-3) / 1e9
            print(f"N={N} B={B} {mode}: {ms:.4f} ms/batch  {B / (ms * 1e-3):.0f} q/s  streamed {hbm:.0f} GB/s ({hbm / 8000:.2f} of HBM)", flush=True)
    del c, m
