"""The main pass of the headline configuration (10M x 384, batch 1024) alone, by device events around the kernel.
With HRS_LIBRARY pointing at the `make noepi` build: the same pass without its epilogue (no results: every query then
takes the exact fallback, so a call takes a second) -- the floor that any work on the epilogue could reach."""
import sys
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0)
N, D, B, K = 10_000_000, 384, 1024, 10
m = torch.empty((N, D), dtype=torch.float32, device='cuda')
for r0 in range(0, N, 1 << 20):
    nr = min(1 << 20, N - r0)
    eng.synth_rows(1235, r0, nr, D, out=m[r0:r0 + nr])
c = eng.corpus(m)
q = eng.synth_rows(4321, 0, B, D)
eng.score_topk(c, q, K, space="cosine", mode="mfma"); torch.cuda.synchronize()
eng.profile(True)
for _ in range(3):
    eng.score_topk(c, q, K, space="cosine", mode="mfma")
torch.cuda.synchronize()
ms, n = eng.profile_read()
print("main pass ms:", ms / n, eng.score_stats(), flush=True)
