"""The fp64 streaming scan over 10M x 384 at one, two and four queries per pass: ms per call and fp32 corpus bytes per
second against the 8 TB/s HBM peak.  usage: python tools/exact_batch.py"""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
D, K, N = 384, 10, 10_000_000
m = torch.empty((N, D), dtype=torch.float32, device='cuda')
for r0 in range(0, N, 1 << 20):
    nr = min(1 << 20, N - r0)
    eng.synth_rows(1235, r0, nr, D, out=m[r0:r0 + nr])
c = eng.corpus(m)
for space in ("l2", "cosine"):
    for B in (1, 2, 4):
        q = eng.synth_rows(4321, 0, B, D)
        eng.score_topk(c, q, K, space=space, mode="exact"); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10): eng.score_topk(c, q, K, space=space, mode="exact")
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 10
        print(f"{space} B={B}: {ms:.3f} ms  {N*D*4/ms/1e6:.0f} GB/s ({N*D*4/ms/1e6/8000:.2f})", flush=True)
