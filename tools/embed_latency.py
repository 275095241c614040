"""Where the time of embedding one short query goes: tokenise, copies, encoder kernels, total."""
import sys, time
sys.path.insert(0, '.')
import torch
from hybrid_rag_system_amd import helper
ef = helper.get_embedding_function()
dev = ef.encoder.engine.device
text = ["How do I restart the nginx container on Unraid?"]
def timed(fn, reps=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
ids, lens = ef.tokenizer.encode_batch(text)
dids, dlens = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
print(f"ids shape {ids.shape}")
print(f"tokenise            {timed(lambda: ef.tokenizer.encode_batch(text)):.3f} ms")
print(f"two host->device    {timed(lambda: (torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev))):.3f} ms")
print(f"encode_ids (queued) {timed(lambda: ef.encoder.encode_ids(dids, dlens)):.3f} ms")
print(f"embed_tensor        {timed(lambda: ef.embed_tensor(text)):.3f} ms")
print(f"embed -> host list  {timed(lambda: ef(text)):.3f} ms")
for B in (8, 64):
    many = text * B
    print(f"embed_tensor x{B:3d}   {timed(lambda: ef.embed_tensor(many), 50):.3f} ms")
