"""Where the int8 matrix-core path overtakes the fp64 streaming scan for 1-4 queries: ms per call of both
over corpora of growing size (squared-L2 top-10)."""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
K = 10
for D in (384, 768):
    for N in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000):
        m = torch.empty((N, D), dtype=torch.float32, device='cuda')
        for r0 in range(0, N, 1 << 20):
            nr = min(1 << 20, N - r0)
            eng.synth_rows(77, r0, nr, D, out=m[r0:r0 + nr])
        c = eng.corpus(m)
        line = f"N={N:8d} D={D} N*D={N*D:.1e}:"
        for B in (1, 4):
            q = eng.synth_rows(78, 0, B, D)
            for mode in ("exact", "mfma"):
                eng.score_topk(c, q, K, mode=mode); torch.cuda.synchronize()
                t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(20): eng.score_topk(c, q, K, mode=mode)
                t1.record(); torch.cuda.synchronize()
                line += f"  B={B} {mode} {t0.elapsed_time(t1)/20:.3f} ms"
        print(line, flush=True)
        c.close(); del m
