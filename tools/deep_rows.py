"""One shard of deep rows (default 4M x 1024, batch 1024, top-100) with the int8 and the bf16 shadow: ms per call
(sequential calls), the main pass by device events, and the call's statistics.
usage: python tools/deep_rows.py [rows [depth [k]]]"""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
B, reps = 1024, 5
eng = Engine(0)
m = torch.empty((N, D), dtype=torch.float32, device='cuda')
for r0 in range(0, N, 1 << 18):
    nr = min(1 << 18, N - r0)
    eng.synth_rows(1235, r0, nr, D, out=m[r0:r0 + nr])
q = eng.synth_rows(4321, 0, B, D)
for kind in ("i8", "bf16"):
    c = eng.corpus(m, shadow=kind)
    eng.score_topk(c, q, K, space="cosine", mode="mfma")
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        eng.score_topk(c, q, K, space="cosine", mode="mfma")
    t1.record()
    torch.cuda.synchronize()
    pr = eng.profile_read()
    eng.profile(False)
    ms = t0.elapsed_time(t1) / reps
    print(f"{kind}@{D} N={N} k={K}: {B / (ms * 1e-3):.0f} q/s  {ms:.3f} ms/call  main pass {pr}  {eng.score_stats()}", flush=True)
    c.close()
