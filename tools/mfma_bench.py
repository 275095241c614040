"""Time the matrix-core scan (main pass) for several shapes: rows x dim x batch."""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
# shapes: NxDxB[xK[xspace[xshadow]]]
shapes = [s.split('x') for s in sys.argv[1:]] or [["1000000", "1024", "1024"], ["10000000", "384", "1024"]]
for sh in shapes:
    n, D, B = int(sh[0]), int(sh[1]), int(sh[2])
    K = int(sh[3]) if len(sh) > 3 else 10
    space = sh[4] if len(sh) > 4 else "l2"
    shadow = sh[5] if len(sh) > 5 else "auto"
    m = torch.empty((n, D), dtype=torch.float32, device='cuda')
    for r0 in range(0, n, 1 << 20):
        nr = min(1 << 20, n - r0)
        eng.synth_rows(1, r0, nr, D, out=m[r0:r0 + nr])
    c = eng.corpus(m, shadow=shadow)
    q = eng.synth_rows(2, 0, B, D)
    eng.score_topk(c, q, K, space=space, mode="mfma"); torch.cuda.synchronize()
    st = eng.score_stats()
    eng.profile(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5): eng.score_topk(c, q, K, space=space, mode="mfma")
    t1.record(); torch.cuda.synchronize()
    ms, k = eng.profile_read(); eng.profile(False)
    Dp = (D + 127) // 128 * 128
    if 512 < Dp < 1024: Dp = 1024
    print(f"N={n} D={D} B={B} k={K} {space} {shadow}: main {ms/k:.3f} ms ({2.0*n*Dp*max(B,128)/(ms/k*1e-3)/1e12:.0f} TF padded)  call {t0.elapsed_time(t1)/5:.3f} ms  {B/(t0.elapsed_time(t1)/5e3):.0f} q/s  {st}", flush=True)
    c.close(); del m
