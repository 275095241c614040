"""Algorithmic GB/s of the exact streaming scan for several row widths (same total bytes)."""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
total = 8 << 30
dims = [int(x) for x in sys.argv[1:]] or [64, 128, 256, 384, 512, 1024]
for D in dims:
    n = total // (4 * D)
    m = torch.empty((n, D), dtype=torch.float32, device='cuda')
    for r0 in range(0, n, 1 << 20):
        nr = min(1 << 20, n - r0)
        eng.synth_rows(1, r0, nr, D, out=m[r0:r0 + nr])
    c = eng.corpus(m)
    for b in (1, 2, 4):
        q = eng.synth_rows(2, 0, b, D)
        eng.score_topk(c, q, 10, mode="exact"); torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(3): eng.score_topk(c, q, 10, mode="exact")
        torch.cuda.synchronize()
        ms, k = eng.profile_read(); eng.profile(False)
        print(f"D={D} B={b}: {ms/k:.3f} ms  {n*D*4/(ms/k*1e-3)/1e9:.0f} GB/s", flush=True)
    c.close(); del m
