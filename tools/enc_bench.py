import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd import weights as W
from hybrid_rag_system_amd.engine import Engine
eng = Engine(0); cfg = W.EncoderConfig(); enc = eng.encoder(W.generate(cfg), cfg)
B, S = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), (int(sys.argv[2]) if len(sys.argv) > 2 else 256)
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.integers(1000, 30000, (B, S)).astype(np.int32)).cuda()
lens = torch.full((B,), S, dtype=torch.int32, device='cuda')
for _ in range(3): out = enc.encode_ids(ids, lens)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): out = enc.encode_ids(ids, lens)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"encode {B}x{S}: {dt*1e3:.3f} ms/batch, {B/dt:.0f} chunks/s, {B*6.040e9/dt/1e12:.1f} TFLOP/s")
