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
