"""The collection store as it grows: cost of an add (1000 rows with given embeddings + the query that follows,
i.e. upload + quantisation of the new rows only) at 10k ... 1M rows, and a `where={"system": ...}` query over
the full collection (mask built on the device from cached value codes)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from hybrid_rag_system_amd.store import Client
target = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(0)
col = Client(None).get_or_create_collection("grow")
D, step = 384, 1000
q = rng.standard_normal((1, D)).astype(np.float32); q /= np.linalg.norm(q)
systems = ["docker", "unraid", "proxmox", "windows", "network"]
marks = {10_000, 100_000, 300_000, target}
have = 0
print("rows      add+query ms (1000 new rows)   query ms   where-query ms")
while have < target:
    emb = rng.standard_normal((step, D)).astype(np.float32); emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    ids = [f"id{have + i}" for i in range(step)]
    metas = [{"system": systems[(have + i) % 5], "n": i} for i in range(step)]
    docs = [f"doc {i}" for i in ids]
    t0 = time.perf_counter()
    col.add(documents=docs, metadatas=metas, ids=ids, embeddings=emb)
    if have + step in marks or (have // step) % 50 == 0:
        col.query(query_embeddings=q, n_results=10); torch.cuda.synchronize()
    t_add = (time.perf_counter() - t0) * 1e3
    have += step
    if have in marks:
        col.query(query_embeddings=q, n_results=10, where={"system": "docker"}); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): col.query(query_embeddings=q, n_results=10)
        torch.cuda.synchronize(); t_q = (time.perf_counter() - t0) * 100
        t0 = time.perf_counter()
        for _ in range(10): r = col.query(query_embeddings=q, n_results=10, where={"system": "docker"})
        torch.cuda.synchronize(); t_w = (time.perf_counter() - t0) * 100
        assert all(m["system"] == "docker" for m in r["metadatas"][0])
        print(f"{have:8d}  {t_add:10.2f}                    {t_q:8.2f}   {t_w:8.2f}", flush=True)
