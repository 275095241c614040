"""Sequential scoring calls at shard sizes (rows x 384, batch 1024, cosine top-10): ms per call and the call's
statistics.  usage: python tools/shard_step.py [rows ...]"""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine

eng = Engine(0)
D, K, B = 384, 10, 1024
for N in [int(a) for a in sys.argv[1:]] or [1_250_000, 2_500_000, 5_000_000]:
    m = torch.empty((N, D), dtype=torch.float32, device='cuda')
    for r0 in range(0, N, 1 << 20):
        nr = min(1 << 20, N - r0)
        eng.synth_rows(1235, r0, nr, D, out=m[r0:r0 + nr])
    c = eng.corpus(m)
    qs = [eng.synth_rows(5000 + i, 0, B, D) for i in range(8)]
    for q in qs:
        eng.score_topk(c, q, K, space="cosine", mode="mfma")
    torch.cuda.synchronize()
    eng.profile(True)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for r in range(5):
        for q in qs:
            eng.score_topk(c, q, K, space="cosine", mode="mfma")
    t1.record()
    torch.cuda.synchronize()
    ms, n = eng.profile_read()
    eng.profile(False)
    print(f"N={N} B={B}: {t0.elapsed_time(t1) / 40:.4f} ms/call  main pass {ms / n:.4f} ms  {eng.score_stats()}", flush=True)
    c.close()
    del m, c
