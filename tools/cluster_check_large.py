"""Clustered corpus at scale (sampling stride 32): exactness, candidate counts, fallbacks, time."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd.engine import Engine
from oracle import search_oracle as so
eng = Engine(0)
torch.manual_seed(3)
d = 384
for csize, noise, n in ((500, 0.3, 5_000_000), (2000, 0.3, 5_000_000), (50, 0.5, 5_000_000)):
    ncl = n // csize
    centers = torch.nn.functional.normalize(torch.randn(ncl, d, device="cuda"), dim=1)
    C = centers.repeat_interleave(csize, dim=0)
    C += (noise / d ** 0.5) * torch.randn_like(C)
    C = torch.nn.functional.normalize(C, dim=1).contiguous()
    B = 1024
    Q = centers[torch.randint(0, ncl, (B,), device="cuda")] + (0.1 / d ** 0.5) * torch.randn(B, d, device="cuda")
    Q = torch.nn.functional.normalize(Q, dim=1).contiguous()
    c = eng.corpus(C)
    eng.score_topk(c, Q, 10, mode="mfma"); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); _, _, keys = eng.score_topk(c, Q, 10, mode="mfma", want_keys=True); t1.record(); torch.cuda.synchronize()
    st = eng.score_stats()
    want = so.topk_keys(Q[:48].cpu().numpy(), C.cpu().numpy(), 10, "l2")
    ok = np.array_equal(keys[:48].cpu().numpy().view(np.uint64), want)
    print(f"cluster {csize} noise {noise} N={n}: exact(48 checked)={ok} call {t0.elapsed_time(t1):.2f} ms {st}", flush=True)
    c.close(); del C
