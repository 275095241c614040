import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
for N, D, K, B in ((4_000_000, 1024, 100, 1024), (4_000_000, 512, 10, 1024)):
    m = torch.empty((N, D), dtype=torch.float32, device='cuda')
    for r0 in range(0, N, 1 << 20):
        nr = min(1 << 20, N - r0)
        eng.synth_rows(1235, r0, nr, D, out=m[r0:r0 + nr])
    c = eng.corpus(m)
    q = eng.synth_rows(4321, 0, B, D)
    for _ in range(3): eng.score_topk(c, q, K, space="cosine", mode="mfma")
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): eng.score_topk(c, q, K, space="cosine", mode="mfma")
    t1.record(); torch.cuda.synchronize()
    ms, n = eng.profile_read(); eng.profile(False)
    print(f"N={N} D={D} k={K}: {t0.elapsed_time(t1)/10:.3f} ms/call, main pass {ms/n:.3f} ms = {2.0*B*N*D/(ms/n*1e-3)/1e12:.0f} TOP/s {eng.score_stats()}", flush=True)
    c.close(); del m, c
