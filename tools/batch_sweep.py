"""10M x 384, squared-L2 top-10, query batch sweep: queries/s, fp32-algorithmic GB/s (the fp32 corpus the
config names, once per batch on the matrix-core path, once per pass of 4 queries on the exact path),
bytes actually streamed, and matrix-core rate, each as a fraction of the MI355X peak."""
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
print("batch  path   ms/batch   queries/s   fp32-algorithmic GB/s   streamed GB/s (frac of 8000)   T(FL)OP/s (frac of peak)")
print("(the matrix-core path streams the int8 shadow, a quarter of the fp32 bytes: its HBM fraction is that of the streamed bytes)")
for B in (1, 2, 4, 8, 32, 128, 256, 512, 1024):
    q = eng.synth_rows(4321, 0, B, D)
    for mode in (("exact", "mfma") if B <= 4 else ("mfma",)) + (("exact",) if B == 8 else ()):
        eng.score_topk(c, q, K, mode=mode); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        reps = 5
        t0.record()
        for _ in range(reps): eng.score_topk(c, q, K, mode=mode)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / reps
        passes = (B + 3) // 4 if mode == "exact" else 1
        alg = N * D * 4 * passes / (ms * 1e-3) / 1e9
        moved = (N * D * 4 * passes if mode == "exact" else N * D) / (ms * 1e-3) / 1e9
        tf = 2.0 * B * N * D / (ms * 1e-3) / 1e12
        peak = 5000.0 if mode == "mfma" else 78.6
        main = ""
        if mode == "mfma":
            # the main pass alone (device events around the kernel): what it reaches of the HBM peak and of the rate an
            # LDS-fed int8 MFMA loop holds on this chip (3050 TOP/s) -- whichever is nearer to 1 bounds it
            eng.profile(True)
            for _ in range(3): eng.score_topk(c, q, K, mode=mode)
            torch.cuda.synchronize()
            pms, pn = eng.profile_read()
            eng.profile(False)
            mp = pms / pn
            main = f"   main pass {mp:.3f} ms: {N * D / (mp * 1e-3) / 1e9 / 8000:.2f} of HBM, {2.0 * B * N * D / (mp * 1e-3) / 1e12 / 3050:.2f} of the measured MFMA rate"
        print(f"{B:5d}  {mode:5s}  {ms:8.3f}  {B/(ms*1e-3):10.0f}   {alg:8.0f}   {moved:8.0f} ({moved/8000:.2f})   {tf:8.1f} ({tf/peak:.2f}){main}", flush=True)
