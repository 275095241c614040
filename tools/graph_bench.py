"""A scoring call captured into a HIP graph (torch.cuda.CUDAGraph around the library's launches on the capture stream):
replay against the plain call, results and time.  usage: python tools/graph_bench.py [N D B K space]"""
import sys, time
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd.engine import Engine
a = sys.argv[1:]
n, D, B, K = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (1_000_000, 384, 256, 10)
space = a[4] if len(a) > 4 else "cosine"
eng = Engine(0)
m = torch.empty((n, D), dtype=torch.float32, device='cuda')
for r0 in range(0, n, 1 << 20):
    nr = min(1 << 20, n - r0)
    eng.synth_rows(1, r0, nr, D, out=m[r0:r0 + nr])
c = eng.corpus(m)
q = eng.synth_rows(2, 0, B, D)
want = eng.score_topk(c, q, K, space=space, mode="mfma", want_keys=True)
torch.cuda.synchronize()
def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
plain = timed(lambda: eng.score_topk(c, q, K, space=space, mode="mfma"))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): eng.score_topk(c, q, K, space=space, mode="mfma")
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g, stream=s):
    out = eng.score_topk(c, q, K, space=space, mode="mfma", want_keys=True)
g.replay(); torch.cuda.synchronize()
assert all(torch.equal(x, y) for x, y in zip(out, want)), "graph replay differs"
q2 = eng.synth_rows(3, 0, B, D)
want2 = eng.score_topk(c, q2, K, space=space, mode="mfma", want_keys=True)
q.copy_(q2); g.replay(); torch.cuda.synchronize()
assert all(torch.equal(x, y) for x, y in zip(out, want2)), "graph replay on new queries differs"
graph = timed(lambda: g.replay())
print(f"N={n} D={D} B={B} k={K} {space}: plain {plain:.3f} ms ({B/plain*1e3:.0f} q/s)   graph replay {graph:.3f} ms ({B/graph*1e3:.0f} q/s)")
