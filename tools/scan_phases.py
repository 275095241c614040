"""Where the time of a small-batch scoring call goes INSIDE the kernels: per-workgroup phases of the main scan pass
and per-query phases of the re-score kernel, from clock reads compiled in by `make -C hybrid_rag_system_amd/csrc prof`
(libhrs_prof.so; the shipped library carries none of this).  The figures are printed to stderr by hrs_score_stats.
usage: HRS_LIBRARY=$PWD/hybrid_rag_system_amd/libhrs_prof.so python tools/scan_phases.py [rows [batch]]"""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
D, K = 384, 10
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 256
m = torch.empty((N, D), dtype=torch.float32, device='cuda')
for r0 in range(0, N, 1 << 20):
    eng.synth_rows(1235, r0, min(1 << 20, N - r0), D, out=m[r0:r0 + min(1 << 20, N - r0)])
c = eng.corpus(m)
for B, space in ((BATCH, "cosine"), (BATCH, "l2")):
    q = eng.synth_rows(4321, 0, B, D)
    for _ in range(3):
        eng.score_topk(c, q, K, space=space, mode="mfma")
    print(B, space, eng.score_stats(), flush=True)
