"""Batches of up to 256 over 10M x 384 on the matrix-core path: call and main-pass time, results checked against the
exact scan on a few queries (run once per library or environment switch under test)."""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
N, D, K = 10_000_000, 384, 10
m = torch.empty((N, D), dtype=torch.float32, device='cuda')
for r0 in range(0, N, 1 << 20):
    nr = min(1 << 20, N - r0)
    eng.synth_rows(1235, r0, nr, D, out=m[r0:r0 + nr])
c = eng.corpus(m)
for B in (int(a) for a in (sys.argv[1:] or ["256"])):
    q = eng.synth_rows(4321, 0, B, D)
    out = eng.score_topk(c, q, K, mode="mfma", want_keys=True); torch.cuda.synchronize()
    nchk = min(B, 8)
    ref = eng.score_topk(c, q[:nchk].contiguous(), K, mode="exact", want_keys=True); torch.cuda.synchronize()
    same = bool(torch.equal(out[2][:nchk], ref[2]))
    for _ in range(3): eng.score_topk(c, q, K, mode="mfma")
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): eng.score_topk(c, q, K, mode="mfma")
    t1.record(); torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(5): eng.score_topk(c, q, K, mode="mfma")
    torch.cuda.synchronize()
    pms, pn = eng.profile_read(); eng.profile(False)
    print(f"B={B}: {t0.elapsed_time(t1) / 10:.3f} ms/call, main pass {pms / pn:.3f} ms, first {nchk} queries equal the exact scan: {same}, {eng.score_stats()}", flush=True)
