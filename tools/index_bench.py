"""End-to-end `index-dir` throughput of the helper: files -> chunks -> tokens -> embeddings -> store."""
import glob, os, sys, tempfile, time, random
sys.path.insert(0, '.')
nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
src = [open(f, encoding="utf-8").read() for f in sorted(glob.glob("tests/golden/context/infrastructure/*.md"))]
rng = random.Random(3)
with tempfile.TemporaryDirectory() as d:
    docs = os.path.join(d, "docs"); os.makedirs(docs)
    os.environ["HRS_DATA_DIR"] = os.path.join(d, "store")
    for i in range(nfiles):
        t = src[i % len(src)]
        paras = t.split("\n\n"); rng.shuffle(paras)
        with open(os.path.join(docs, f"doc{i:05d}.md"), "w", encoding="utf-8") as f:
            f.write(f"# document {i}\n\n" + "\n\n".join(paras))
    from hybrid_rag_system_amd import helper
    helper.embed(["warm up"])  # loads the library, builds the encoder
    import torch; torch.cuda.synchronize()
    t0 = time.time()
    r = helper.index_directory(docs, collection_name="bench")
    torch.cuda.synchronize()
    dt = time.time() - t0
    chunks = sum(f["chunks"] for f in r["files"])
    print(f"{r['indexed']} files, {chunks} chunks in {dt:.2f} s: {chunks/dt:.0f} chunks/s")
    t0 = time.time()
    out = helper.hybrid_query_batch(["proxmox backup schedule", "docker network bridge"] * 64, n_results=5, collection_name="bench")
    print(f"128 hybrid queries: {(time.time()-t0)*1e3:.1f} ms")
    # one query at a time, as the reference's CLI and bridge issue them (in-process: without the interpreter start)
    for name, fn in (("embed", lambda: helper.embed(["proxmox backup schedule"])),
                     ("query", lambda: helper.query("proxmox backup schedule", 5, "bench")),
                     ("hybrid_query", lambda: helper.hybrid_query("proxmox backup schedule", 5, "bench"))):
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(50): fn()
        torch.cuda.synchronize()
        print(f"single {name}: {(time.time()-t0)/50*1e3:.2f} ms")
