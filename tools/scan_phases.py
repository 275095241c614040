"""Where the time of a small-batch scoring call goes INSIDE the kernels: per-workgroup phases of the main scan pass
and per-query phases of the re-score kernel, from clock reads compiled in by `make -C hybrid_rag_system_amd/csrc prof`
(libhrs_prof.so; the shipped library carries none of this).  The figures are printed to stderr by hrs_score_stats.
usage: HRS_LIBRARY=$PWD/hybrid_rag_system_amd/libhrs_prof.so python tools/scan_phases.py"""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
D, K = 384, 10
N = 1_000_000
m = torch.empty((N, D), dtype=torch.float32, device='cuda')
eng.synth_rows(1235, 0, N, D, out=m)
c = eng.corpus(m)
for B, space in ((256, "cosine"), (256, "l2")):
    q = eng.synth_rows(4321, 0, B, D)
    for _ in range(3):
        eng.score_topk(c, q, K, space=space, mode="mfma")
    print(B, space, eng.score_stats(), flush=True)
