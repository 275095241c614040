"""Does the encoder gain from two half-batches on two HIP streams (memory-bound kernels of one half beside the
matrix-bound kernels of the other)?  python tools/enc_two_streams.py [B] -- prints one-stream and two-stream rates."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd import weights as W
from hybrid_rag_system_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = 256
eng = Engine(0); cfg = W.EncoderConfig(); wts = W.generate(cfg)
encs = [eng.encoder(wts, cfg), eng.encoder(wts, cfg)]   # one workspace each
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.integers(1000, 30000, (B, S)).astype(np.int32)).cuda()
lens = torch.full((B,), S, dtype=torch.int32, device='cuda')
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
h = B // 2
halves = [(ids[:h].contiguous(), lens[:h].contiguous()), (ids[h:].contiguous(), lens[h:].contiguous())]

def one():
    return encs[0].encode_ids(ids, lens)

def two():
    outs = []
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            outs.append(encs[i].encode_ids(*halves[i]))
    return outs

ref = one(); torch.cuda.synchronize()
o2 = two(); torch.cuda.synchronize()
print("two-stream result equal to one-stream:", bool(torch.equal(torch.cat(o2), ref)))
for name, f in (("one stream", one), ("two streams", two)):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt*1e3:.3f} ms per {B} x {S} batch, {B/dt:.0f} chunks/s")
