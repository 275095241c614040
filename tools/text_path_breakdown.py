"""Host-side stages of the text query path for 1024 short queries, timed apart (keyword extraction, term ids,
tokenisation), and the GPU stages with a synchronisation behind each.  usage: python tools/text_path_breakdown.py"""
import sys, time, zlib
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd import textproc
from hybrid_rag_system_amd.engine import Engine
import hybrid_rag_system_amd.store as store
from hybrid_rag_system_amd.store import HrsEmbeddingFunction

eng = Engine(0)
store._ENGINE = eng
ef = HrsEmbeddingFunction()
rng = np.random.default_rng(11)
words = ["restart", "docker", "container", "unraid", "backup", "schedule", "proxmox", "firewall", "address", "powershell", "filter",
         "network", "storage", "array", "parity", "service", "status", "failed", "certificate", "renewal", "tunnel", "forwarding",
         "how", "do", "the", "on", "with", "what", "is", "of"]
B = 1024
texts = [" ".join(rng.choice(words, int(rng.integers(5, 12)))) for _ in range(B)]

def t(f, n=5):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r

ms, kws = t(lambda: [textproc.extract_keywords(x) for x in texts])
print(f"extract_keywords x{B}: {ms:.3f} ms")
ms, _ = t(lambda: np.fromiter((zlib.crc32(w.encode()) % 50000 for k in kws for w in k), np.int32))
print(f"term ids (crc32): {ms:.3f} ms")
order = sorted(range(B), key=lambda i: len(texts[i]))
ms, (ids, lens) = t(lambda: ef.tokenizer.encode_batch([texts[i] for i in order]))
print(f"sort + tokenise: {ms:.3f} ms  ids {ids.shape}")
ms, _ = t(lambda: (torch.from_numpy(ids).to(eng.device), torch.from_numpy(lens).to(eng.device)))
print(f"upload ids: {ms:.3f} ms")
di, dl = torch.from_numpy(ids).to(eng.device), torch.from_numpy(lens).to(eng.device)
ms, _ = t(lambda: ef.encoder.encode_ids(di, dl))
print(f"encode (GPU, synchronised): {ms:.3f} ms")
ms, _ = t(lambda: ef.embed_tensor(texts))
print(f"embed_tensor whole: {ms:.3f} ms")
