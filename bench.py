#!/usr/bin/env python3
"""Headline benchmark: hybrid top-10 queries/sec over a 10M x 384 corpus (BASELINE.json config 4).

One step = one pass of the hot path over one batch of 1024 queries whose embeddings are already in
HBM: exact top-20 by squared-L2 distance (int8 MFMA scan + exact fp64 re-score), keyword presence
counts of those 20 rows over forward-CSR postings, fused score 0.7*sem + 0.3*kw, stable order, top-10.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With N > 1 the headline is the row-shard layout: the corpus rows (and their term lists) are split over the
ranks, every rank scores the same batch against its shard, and every batch ends with ONE all-gather of the
per-shard candidates (RCCL over xGMI: B x 2n x 16 bytes per rank) and a merge on every rank -- strong scaling
(total work fixed).  The replica layout (every rank holds the whole corpus and answers its own batch; no
data-path collective, weak scaling) is measured after it and reported under "extras" when the corpus fits
one GPU (--layout replica makes it the headline instead).  Rank 0 prints ONE JSON line.

The 100M x 1024 configuration (BASELINE.json config 5; 12.5M rows = 51 GB fp32 per GPU, top-100 candidates):
  python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --rows 100000000 --dim 1024 --n-results 50 --space cosine
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hybrid_rag_system_amd.dist import HybridPipeline, TopKPipeline, shard_bounds  # noqa: E402
from hybrid_rag_system_amd.engine import Engine  # noqa: E402

SEED_CORPUS, SEED_QUERY, SEED_TERMS, SEED_POST = 1235, 4321, 777, 99
VOCAB, TERMS_PER_DOC = 50_000, 96
MFMA_PEAK_TFLOPS = {"i8": 5000.0, "bf16": 2500.0}  # dense matrix-core peaks of MI355X (int8 counts integer ops)
HBM_PEAK_GBPS = 8000.0
# what this chip has been measured to sustain (for the "of measured" fractions beside the spec ones): a plain copy
# kernel (the MI355X guide's figure) and an LDS-fed MFMA loop on random operands (profiles/r03_mfma_micro.txt)
HBM_MEASURED_GBPS = 6290.0
MFMA_MEASURED_TFLOPS = {"i8": 3050.0, "bf16": 1320.0}  # LDS-fed loops on random operands: int8 16x16x64 without barrier or DMA 3011-3080, bf16 16x16x32 1296-1348


def zipf_cdf(V, device):
    w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
    cdf = torch.cumsum(w, 0)
    return (cdf / cdf[-1]).float().to(device)


def make_queries(eng, B, D, n_total, stream=0):
    """Unit queries; every 10th is a corpus row plus N(0, 0.05^2/D)-scaled noise, so max(0, 1-d) > 0.
    `stream` selects an independent batch (one per rank in the replica layout)."""
    q = eng.synth_rows(SEED_QUERY + 1000 * stream, 0, B, D)
    rng = np.random.default_rng(SEED_QUERY + 1000 * stream)
    planted = rng.integers(0, n_total, B // 10)
    noise = eng.synth_rows(SEED_QUERY + 1000 * stream + 1, 0, B // 10, D, normalize=False)
    for j, r in enumerate(planted.tolist()):
        row = eng.synth_rows(SEED_CORPUS, r, 1, D)  # any rank can regenerate any global row
        v = row[0] + 0.05 * noise[j]
        q[j * 10] = v / v.norm()
    return q.contiguous()


def make_query_terms(B, device):
    rng = np.random.default_rng(SEED_TERMS)
    p = 1.0 / np.arange(1, VOCAB + 1)
    p /= p.sum()
    m = rng.integers(1, 9, B)
    terms = [rng.choice(VOCAB, int(mm), replace=False, p=p) for mm in m]
    offs = np.concatenate([[0], np.cumsum(m)]).astype(np.int32)
    return torch.from_numpy(np.concatenate(terms).astype(np.int32)).to(device), torch.from_numpy(offs).to(device)


def cpu_baseline(corpus_matrix, terms, queries, q_terms, q_offs, n_keep, n_total, sample_rows, sample_queries):
    """The CPU oracle (a port, not the reference: ChromaDB cannot run here) on a bounded sample."""
    from oracle import search_oracle as so

    ns = min(sample_rows, corpus_matrix.shape[0])
    bs = min(sample_queries, queries.shape[0])
    C = corpus_matrix[:ns].cpu().numpy()
    T = terms[:ns].cpu().numpy().reshape(-1)
    offs = np.arange(ns + 1, dtype=np.int64) * terms.shape[1]
    Q = queries[:bs].cpu().numpy()
    qo = q_offs[: bs + 1].cpu().numpy()
    qt = q_terms[: int(qo[-1])].cpu().numpy()
    so.lib()
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    t0 = time.perf_counter()
    keys = so.topk_keys(Q, C, 2 * n_keep, "l2")
    d, r = so.unpack(keys)
    cnt = so.keyword_counts(offs, T, qo, qt, r)
    so.fuse_sort(d, cnt, np.diff(qo), n_keep, 0.3)
    dt = time.perf_counter() - t0
    qps_full = bs / dt * (ns / n_total)  # linear in corpus rows
    # beside it, what a CPU vector store would do at best: one fp32 sgemm (all host threads) + top-k; not
    # bit-comparable with the oracle (fp32 accumulation), reported for scale only
    tq, tc = torch.from_numpy(Q), torch.from_numpy(C)
    (tq @ tc[:1000].T)
    t0 = time.perf_counter()
    torch.topk(2.0 - 2.0 * (tq @ tc.T), 2 * n_keep, dim=1, largest=False)
    dts = time.perf_counter() - t0
    return {
        "value": qps_full,
        "unit": "queries/s",
        "cores": min(threads, bs),
        "kind": "port",
        "sample": f"oracle/search_oracle.c on {bs} queries x {ns} rows x {C.shape[1]} dims in {dt:.2f}s, scaled linearly to {n_total} rows",
        "sgemm_topk_queries_per_s": bs / dts * (ns / n_total),
        "sgemm_threads": torch.get_num_threads(),
    }


def bench_embed(eng, batches=4, B=1024, S=256, cpu_chunks=2048):
    """BASELINE config 3 shape: MiniLM-L6 embedding of full-length (256-token) synthetic chunks, batch 1024;
    beside it the fp32 oracle (torch on the host cores, sentence-transformers' default batch of 32)."""
    from hybrid_rag_system_amd import weights as W

    cfg = W.EncoderConfig()
    wts = W.generate(cfg, 2026)
    enc = eng.encoder(wts, cfg)
    g = torch.Generator(device="cpu").manual_seed(3)
    ids = torch.randint(1000, 30000, (B, S), generator=g, dtype=torch.int32)
    ids[:, 0], ids[:, -1] = 101, 102
    ids = ids.to(eng.device)
    lens = torch.full((B,), S, dtype=torch.int32, device=eng.device)
    for _ in range(2):
        enc.encode_ids(ids, lens)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(batches):
        enc.encode_ids(ids, lens)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / batches
    # the same number of chunks with lengths ~U[32, 256], sorted by length into batches (what the store's
    # embedding function does), each batch padded to its longest chunk rounded up to 32 tokens
    gl = torch.Generator(device="cpu").manual_seed(4)
    rl, _ = torch.sort(torch.randint(32, S + 1, (batches * B,), generator=gl, dtype=torch.int32))
    ragged = []
    for b in range(batches):
        bl = rl[b * B : (b + 1) * B]
        sb = (int(bl.max()) + 31) // 32 * 32
        bi = torch.randint(1000, 30000, (B, sb), generator=gl, dtype=torch.int32)
        bi[:, 0] = 101
        bi[torch.arange(B), (bl - 1).long()] = 102
        ragged.append((bi.to(eng.device), bl.to(eng.device)))
    for bi, bl in ragged:
        enc.encode_ids(bi, bl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for bi, bl in ragged:
        enc.encode_ids(bi, bl)
    torch.cuda.synchronize()
    rdt = time.perf_counter() - t0
    real_tokens = int(rl.sum())
    enc.close()
    flops = 6.040e9 * B  # 23 592 960 flop per token x 256 tokens
    out = {"chunks_per_s": B / dt, "ms_per_batch": dt * 1e3, "tflops": flops / dt / 1e12, "mfma_frac": flops / dt / 1e12 / MFMA_PEAK_TFLOPS["bf16"],
           "config": f"MiniLM-L6 geometry, generated weights, batch {B} x {S} tokens, bf16 MFMA",
           "ragged": {"chunks_per_s": batches * B / rdt, "real_tokens_per_s": real_tokens / rdt, "mean_len": real_tokens / (batches * B),
                      "config": f"{batches * B} chunks, lengths ~U[32,{S}], length-sorted batches of {B}"}}
    if cpu_chunks:
        from oracle import encoder_oracle as EO  # the checker, timed as the CPU baseline of this extra

        # sentence-transformers' default on the reference's CPU path: batches of 32
        reps = (cpu_chunks + B - 1) // B
        hid = ids.cpu().numpy()
        hl = lens.cpu().numpy()
        EO.forward(wts, hid[:4], hl[:4], cfg)
        t0 = time.perf_counter()
        done = 0
        for r in range(reps):
            for s0 in range(0, B, 32):
                if done >= cpu_chunks:
                    break
                EO.forward(wts, hid[s0 : s0 + 32], hl[s0 : s0 + 32], cfg)
                done += 32
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": done / cdt, "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"oracle/encoder_oracle.py (fp32 torch) on {done} chunks x {S} tokens in batches of 32 in {cdt:.2f}s"}
    return out


def bench_scan(eng, corpus, queries, k=10):
    """The small-batch streaming scan (fp64-exact, fp32 corpus read once per pass of <= 4 queries)."""
    out = {}
    for b in (1, 4):
        q = queries[:b].contiguous()
        eng.score_topk(corpus, q, k, mode="exact")
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(5):
            eng.score_topk(corpus, q, k, mode="exact")
        torch.cuda.synchronize()
        ms, n = eng.profile_read()
        eng.profile(False)
        gbps = corpus.n * corpus.dim * 4 / (ms / n * 1e-3) / 1e9
        out[f"B{b}"] = {"kernel_ms": ms / n, "algorithmic_GBps": gbps, "hbm_frac": gbps / HBM_PEAK_GBPS, "queries_per_s": b / (ms / n * 1e-3)}
    return out


def bench_bm25_step(eng, corpus, terms, inv, queries, q_terms, q_offs, n_keep, k1=1.2, b=0.75, engines=None):
    """The headline's hybrid step with BM25 weights in the place of the keyword fraction (an extension: the reference
    scores the fraction of keywords present): top-2n by distance, BM25 of the candidates from their term lists (term
    frequency, list length, idf from the document frequencies of the inverted postings), fusion, first n.  Checked on
    sampled queries against the same arithmetic in Python floats, bit for bit (the run aborts otherwise)."""
    from hybrid_rag_system_amd.engine import Postings

    n = corpus.n
    B = queries.shape[0]
    df = (inv.term_offs[1:] - inv.term_offs[:-1]).to(torch.float64)
    idf = torch.log(1.0 + (float(n) - df + 0.5) / (df + 0.5)).contiguous()
    doc_offs = torch.arange(n + 1, dtype=torch.int64, device=eng.device) * TERMS_PER_DOC
    postings = Postings(eng, doc_offs, terms.reshape(-1), 0)
    par = {"idf": idf, "avg_len": float(TERMS_PER_DOC), "k1": k1, "b": b}
    run = lambda: eng.hybrid_topk(corpus, postings, queries, q_terms, q_offs, n_keep, bm25=par)
    out = run()
    torch.cuda.synchronize()
    idf_h, qt_h, qo_h = idf.cpu().numpy(), q_terms.cpu().numpy(), q_offs.cpu().numpy()
    for q in (0, B // 2, B - 1):
        rows, dist = out["cand_rows"][q].cpu().numpy(), out["cand_dist"][q].cpu().numpy()
        kw_gpu = out["cand_kw"][q].cpu().numpy()
        mine = [int(t) for t in qt_h[qo_h[q] : qo_h[q + 1]]]
        scored = []
        for j, r in enumerate(rows):
            lst = terms[int(r)].cpu().numpy().tolist()
            norm = k1 * ((1.0 - b) + b * (float(len(lst)) / float(TERMS_PER_DOC)))
            score = best = 0.0
            for t in mine:
                w = float(idf_h[t])
                best = best + w * (k1 + 1.0)
                f = lst.count(t)
                if f > 0:
                    score = score + ((w * float(f)) * (k1 + 1.0)) / (float(f) + norm)
            kw = score / best if best > 0.0 else 0.0
            if np.float64(kw).view(np.uint64) != kw_gpu[j : j + 1].view(np.uint64)[0]:
                raise SystemExit(f"BM25 hybrid step: keyword weight of query {q}, candidate {j} differs from the host recount")
            scored.append((max(0.0, 1.0 - float(dist[j])) * (1.0 - 0.3) + kw * 0.3, j))
        scored.sort(key=lambda x: x[0], reverse=True)
        want = [j for _, j in scored[:n_keep]]
        if out["perm"][q].cpu().numpy().tolist() != want:
            raise SystemExit(f"BM25 hybrid step: fused order of query {q} differs from the host's")
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out = {"queries_per_s": B / dt, "ms_per_batch": dt * 1e3, "batch": B,
           "config": f"sequential calls: top-{2 * n_keep} by distance over {n} x {corpus.dim}, BM25 (k1 {k1}, b {b}) of the candidates over their term lists, fusion -> top-{n_keep}",
           "checked": "3 queries: BM25 weights and fused order equal Python-float arithmetic bit for bit"}
    if engines is not None and len(engines) > 1:
        # the same step through the two-stream pipeline of the headline (the scan of batch i + 1 beside the tail of batch i)
        pipe = HybridPipeline(engines, corpus, postings, group=False)
        step = lambda: pipe.submit(queries, q_terms, q_offs, n_keep, bm25=par)
        for _ in range(4):
            step()
        pipe.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        last = pipe.drain()
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / reps
        if not (torch.equal(last["perm"], run()["perm"])):
            raise SystemExit("BM25 hybrid step: the pipelined batch differs from the single call")
        out["pipelined_ms_per_batch"] = dtp * 1e3
        out["pipelined_queries_per_s"] = B / dtp
    postings.close()
    return out


def bench_full_corpus(eng, corpus, terms, queries, q_terms, q_offs, n_keep, B=64, engines=None):
    """The full-corpus hybrid mode (an extension: the fused score over EVERY row, not over the 2n nearest) at the
    headline's size: match counts of all rows from inverted postings, the nearest n of every match-count class -- all
    classes of four queries in one fp64 pass over the fp32 corpus -- and the fusion.  Checked on the first query: the
    class of every returned row equals a host recount from the row's term list, and no row of a higher class than the
    winner's lies nearer than the winner (the run aborts otherwise)."""
    from hybrid_rag_system_amd.engine import InvertedPostings

    n = corpus.n
    t0 = time.perf_counter()
    doc_offs = torch.arange(n + 1, dtype=torch.int64, device=eng.device) * TERMS_PER_DOC
    inv = InvertedPostings.from_forward(eng, doc_offs, terms.reshape(-1), VOCAB)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    q = queries[:B].contiguous()
    qo = q_offs[: B + 1].contiguous()
    qt = q_terms[: int(qo[-1].item())].contiguous()
    run = lambda: eng.hybrid_full_corpus(corpus, inv, q, qt, qo, n_keep, 0.3, "l2")
    out = run()
    torch.cuda.synchronize()
    rows0, cnt0 = out["rows"][0].cpu().numpy(), out["counts"][0].cpu().numpy()
    mine = set(qt[: int(qo[1].item())].cpu().numpy().tolist())
    recount = [len(mine.intersection(terms[r].cpu().numpy().tolist())) for r in rows0 if r >= 0]
    if recount != cnt0[: len(recount)].tolist() or not (np.diff(out["combined"][0].cpu().numpy()) <= 0).all():
        raise SystemExit("full-corpus hybrid: match counts of the returned rows differ from a host recount")
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    levels = int((qo[1:] - qo[:-1]).max().item()) + 1
    bm25 = bench_bm25_step(eng, corpus, terms, inv, queries, q_terms, q_offs, n_keep, engines=engines)
    inv.close()
    passes = (B + 3) // 4
    return {"bm25_hybrid_step": bm25, "queries_per_s": B / dt, "ms_per_batch": dt * 1e3, "batch": B, "classes": levels, "corpus_passes": passes,
            "algorithmic_GBps": passes * n * corpus.dim * 4 / dt / 1e9, "hbm_frac": passes * n * corpus.dim * 4 / dt / 1e9 / HBM_PEAK_GBPS,
            "inverted_index_build_s": build_s,
            "config": f"{n} x {corpus.dim} fp32 corpus, {B} queries of 1-8 keywords, top-{n_keep} of the whole corpus under the fused score; the time includes the match counts of every row and the fusion",
            "checked": "first query: classes of the returned rows equal a host recount; fused scores descend"}


def bench_other_configs(eng, eng2):
    """BASELINE configs 2 and 5 (one shard) on one GPU: plain exact top-k scoring (no keyword stage), MFMA path.
    Checked figures: the exact fallback must stay the exception and a sample of the batch must equal the fp64
    streaming scan's keys bit for bit, or the run aborts."""
    out = {}
    for name, n, D, B, k in (("config2_1Mx384_cosine_top10_batch256", 1_000_000, 384, 256, 10),
                             ("config5_one_shard_12.5Mx1024_cosine_top100_batch1024", 12_500_000, 1024, 1024, 100)):
        m = torch.empty((n, D), dtype=torch.float32, device=eng.device)
        for r0 in range(0, n, 1 << 20):
            nr = min(1 << 20, n - r0)
            eng.synth_rows(SEED_CORPUS + 5, r0, nr, D, out=m[r0 : r0 + nr])
        c = eng.corpus(m)
        q = eng.synth_rows(SEED_QUERY + 5, 0, B, D)
        _, _, keys = eng.score_topk(c, q, k, space="cosine", mode="mfma", want_keys=True)
        torch.cuda.synchronize()
        st = eng.score_stats()
        if st["fallback_queries"] > B // 100 or st["overflow_queries"]:
            raise SystemExit(f"{name}: {st} -- the matrix-core path did not prove its results")
        sel = torch.tensor([0, 1, B // 2, B - 1], device=eng.device)
        _, _, kx = eng.score_topk(c, q[sel].contiguous(), k, space="cosine", mode="exact", want_keys=True)
        if not torch.equal(keys[sel], kx):
            raise SystemExit(f"{name}: matrix-core keys differ from the exact scan's")
        reps = 200 if n <= 2_000_000 else 10  # (a 0.2 ms call: enough of them that the final synchronisation does not show)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.score_topk(c, q, k, space="cosine", mode="mfma")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        # the main pass alone, from an event pair around it, in calls of their own (the events cost a call of
        # 0.2 ms a tenth of its time)
        eng.profile(True)
        for _ in range(5):
            eng.score_topk(c, q, k, space="cosine", mode="mfma")
        torch.cuda.synchronize()
        ms, nl = eng.profile_read()
        eng.profile(False)
        # steady-state throughput: consecutive batches alternate over two contexts / streams (the tail of a batch
        # under the next batch's scan), as the headline step does
        pipe = TopKPipeline([eng, eng2], c)
        for _ in range(4):
            pipe.submit(q, k, space="cosine", mode="mfma")
        pipe.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2 * reps):
            pipe.submit(q, k, space="cosine", mode="mfma")
        last = pipe.drain()
        torch.cuda.synchronize()
        pdt = (time.perf_counter() - t0) / (2 * reps)
        if not torch.equal(last[2], keys):
            raise SystemExit(f"{name}: the pipelined call returned other keys")
        eng.set_peer(None)
        eng2.set_peer(None)
        out[name] = {"ms_per_batch": dt * 1e3, "queries_per_s": B / dt, "pipelined_ms_per_batch": pdt * 1e3, "pipelined_queries_per_s": B / pdt,
                     "scan_kernel_ms": ms / max(nl, 1), "scan_share_of_pipelined_step": ms / max(nl, 1) / (pdt * 1e3),
                     "scan_TFLOPs": 2.0 * n * D * B / (ms / max(nl, 1) * 1e-3) / 1e12, "fallback_queries": st["fallback_queries"],
                     "checked": "4 sampled queries equal the exact scan's keys; the pipelined batch equals the single call"}
        c.close()
        del m, c
        torch.cuda.empty_cache()
    return out


def bench_text_queries(eng, searcher, n_keep, B=1024, reps=5):
    """End to end from TEXT: tokenise B short queries (host threads), encode them (MiniLM geometry, generated
    weights), extract keywords (host), then the hybrid step over the resident corpus."""
    from hybrid_rag_system_amd import textproc
    from hybrid_rag_system_amd.store import HrsEmbeddingFunction
    import hybrid_rag_system_amd.store as store

    store._ENGINE = eng
    ef = HrsEmbeddingFunction()
    rng = np.random.default_rng(11)
    words = ["restart", "docker", "container", "unraid", "backup", "schedule", "proxmox", "firewall", "address", "powershell", "filter",
             "network", "storage", "array", "parity", "service", "status", "failed", "certificate", "renewal", "tunnel", "forwarding",
             "how", "do", "the", "on", "with", "what", "is", "of"]
    texts = [" ".join(rng.choice(words, int(rng.integers(5, 12)))) for _ in range(B)]

    def once():
        kws = [textproc.extract_keywords(t) for t in texts]
        offs = np.zeros(B + 1, np.int32)
        offs[1:] = np.cumsum([len(k) for k in kws])
        terms = np.fromiter((zlib.crc32(w.encode()) % VOCAB for k in kws for w in k), np.int32, int(offs[-1]))
        q = ef.embed_tensor(texts)
        return searcher.step(q.contiguous(), torch.from_numpy(terms).to(eng.device), torch.from_numpy(offs).to(eng.device), n_keep)

    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ef.encoder.close()
    return {"queries_per_s": B / dt, "ms_per_batch": dt * 1e3, "config": f"{B} text queries of 5-11 words: host tokeniser + keyword extraction, GPU encoder, hybrid top-{n_keep}"}


def build_rank_state(eng, dev, n_total, D, lo, hi, shadow):
    """This rank's rows [lo, hi) of the synthetic corpus, its shadow and its postings."""
    n_local = hi - lo
    matrix = torch.empty((n_local, D), dtype=torch.float32, device=dev)
    step_rows = 1 << 20
    for r0 in range(0, n_local, step_rows):
        nr = min(step_rows, n_local - r0)
        eng.synth_rows(SEED_CORPUS, lo + r0, nr, D, out=matrix[r0 : r0 + nr])
    corpus = eng.corpus(matrix, row_offset=lo, shadow=shadow)
    cdf = zipf_cdf(VOCAB, dev)
    terms = eng.synth_postings(SEED_POST, lo, n_local, cdf, TERMS_PER_DOC)
    doc_offs = torch.arange(n_local + 1, dtype=torch.int64, device=dev) * TERMS_PER_DOC
    postings = eng.postings(doc_offs, terms.reshape(-1), row_offset=lo)
    return matrix, corpus, terms, postings


def check_step(eng, corpus, terms, lo, queries, q_terms, q_offs, out, n_keep, space, sharded, sample=24):
    """What the timed steps computed, checked on a sample of the batch at full size (the run aborts otherwise):
    the candidates equal the fp64 streaming scan's (single GPU), their keyword counts equal a recount on the host
    from the rows' own term lists, and order and fused scores equal the helper's arithmetic
    (scripts/chromadb_helper.py:302-325) in Python floats, bit for bit."""
    B, c = out["cand_rows"].shape
    sel = sorted(set(np.linspace(0, B - 1, sample).astype(int).tolist()) | {0, 10, 20})  # every 10th query is a planted one
    sel_t = torch.tensor(sel, device=queries.device)
    rows = out["cand_rows"][sel_t].cpu().numpy()
    dist = out["cand_dist"][sel_t].cpu().numpy()
    counts = out["cand_counts"][sel_t].cpu().numpy()
    comb = out["combined"][sel_t].cpu().numpy()
    perm = out["perm"][sel_t].cpu().numpy()
    if not sharded:
        xd, xr = eng.score_topk(corpus, queries[sel_t].contiguous(), c, space=space, mode="exact")
        if not (torch.equal(xr.cpu(), torch.from_numpy(rows)) and np.array_equal(xd.cpu().numpy().view(np.uint32), dist.view(np.uint32))):
            raise SystemExit("headline check: the candidates differ from the exact scan's")
    qo = q_offs.cpu().numpy()
    qt = q_terms.cpu().numpy()
    n_local = terms.shape[0]
    for i, b in enumerate(sel):
        mine = qt[qo[b] : qo[b + 1]].tolist()
        m = len(mine)
        local = (rows[i] >= lo) & (rows[i] < lo + n_local)
        if local.any():
            rt = terms[torch.from_numpy(rows[i][local] - lo).to(terms.device)].cpu().numpy()
            want = np.array([sum(1 for t in mine if t in set(r.tolist())) for r in rt])
            if not np.array_equal(want, counts[i][local]):
                raise SystemExit(f"headline check: keyword counts of query {b} differ from the recount")
        scored = []
        for j in range(c):
            if rows[i][j] < 0:
                continue
            d = float(dist[i][j])
            sem = max(0, 1 - d)
            kw = int(counts[i][j]) / m if m else 0.0
            scored.append((sem * (1 - 0.3) + kw * 0.3, j))
        scored.sort(key=lambda x: x[0], reverse=True)  # stable, as list.sort in the helper
        scored = scored[:n_keep]
        if [j for _, j in scored] != perm[i][: len(scored)].tolist() or not np.array_equal(np.array([v for v, _ in scored], np.float64).view(np.uint64), comb[i][: len(scored)].view(np.uint64)):
            raise SystemExit(f"headline check: order or fused scores of query {b} differ from the helper's arithmetic")
    return f"{len(sel)} sampled queries of the last timed batch: candidates equal the fp64 exact scan's" + ("" if not sharded else " (skipped: sharded)") + ", keyword counts equal a host recount from the rows' term lists, order and fused scores equal Python-float arithmetic bit for bit"


def timed_run(engines, corpus, postings, group, queries, q_terms, q_offs, n_keep, args, world, rehearsal, dev):
    """W warmup + K timed steps through the two-stream pipeline -> (max-over-ranks seconds, kernel ms, launches, stats, pipeline, last result)."""
    import torch.distributed as dist

    eng = engines[0]
    searcher = HybridPipeline(engines, corpus, postings, group=group)

    def step():
        return searcher.submit(queries, q_terms, q_offs, n_keep, mode=args.mode, space=args.space)

    for _ in range(len(engines)):  # every engine sizes its workspaces before the warmup proper
        step()
    searcher.drain()
    for _ in range(args.warmup):
        step()
    searcher.drain()
    torch.cuda.synchronize()
    stats = eng.score_stats()

    def barrier():
        if world > 1:
            dist.barrier()

    eng.profile(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    last = searcher.drain()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = eng.profile_read()
    eng.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, kern_ms, launches, stats, searcher, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    # (the chip's clock settles over the first eight or so main passes of a run: 3.1 -> 2.75 ms in a kernel trace)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n-results", type=int, default=10)
    ap.add_argument("--space", default="l2", choices=["l2", "cosine", "ip"], help="distance of the collection (the reference's default space is l2)")
    ap.add_argument("--mode", default="auto", choices=["auto", "exact", "mfma"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample-queries", type=int, default=1024)
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2, 3], help="consecutive batches alternate over this many HIP streams (1: strictly back to back)")
    ap.add_argument("--layout", default="auto", choices=["auto", "replica", "shard"], help="multi-GPU layout of the headline (auto: shard -- the rows split over the ranks, one all-gather per batch)")
    ap.add_argument("--shadow", default="auto", choices=["auto", "i8", "bf16"], help="scan shadow of the corpus (auto: int8)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (embed, small-batch scan, other configs, text queries, replica layout)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # rehearsal on a one-GPU box: every rank on device 0, gloo instead of RCCL (which refuses two ranks on
    # one device); never the measured configuration
    rehearsal = os.environ.get("HRS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        import faulthandler

        local_rank = 0
        faulthandler.dump_traceback_later(300, exit=True)  # a rehearsal that stops says where
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        backend = dist.get_backend()
    eng = Engine(local_rank)

    n_total, D, B, n_keep = args.rows, args.dim, args.batch, args.n_results
    footprint = n_total * (D * 4 + D * 2 + TERMS_PER_DOC * 4 + 8)  # fp32 rows + largest shadow + postings
    layout = "single" if world == 1 else ("shard" if args.layout == "auto" else args.layout)
    if layout == "replica" and footprint > 200e9:
        raise SystemExit("--layout replica: the corpus does not fit one GPU")
    lo, hi = shard_bounds(n_total, world, rank) if layout == "shard" else (0, n_total)
    n_local = hi - lo
    matrix, corpus, terms, postings = build_rank_state(eng, dev, n_total, D, lo, hi, args.shadow)
    shadow = args.shadow if args.shadow != "auto" else "i8"
    queries = make_queries(eng, B, D, n_total, stream=rank if layout == "replica" else 0)
    q_terms, q_offs = make_query_terms(B, dev)
    torch.cuda.synchronize()

    # batch i runs on stream i % streams (one engine = one set of workspaces per stream); the dominant
    # kernel is timed on the first engine's launches
    engines = [eng] + [Engine(local_rank) for _ in range(max(args.streams, 1) - 1)]
    elapsed, kern_ms, launches, stats, searcher, last = timed_run(engines, corpus, postings, None if layout == "shard" else False, queries, q_terms,
                                                                  q_offs, n_keep, args, world, rehearsal, dev)
    bytes_gathered = searcher.lanes[0][0].bytes_gathered
    checked = check_step(eng, corpus, terms, lo, queries, q_terms, q_offs, last, n_keep, args.space, layout == "shard" and world > 1)

    # the replica layout beside a sharded headline (when the whole corpus fits one GPU next to the shard)
    replica = None
    if layout == "shard" and not args.no_extras and footprint <= 96e9:
        corpus.close()
        postings.close()
        del matrix, corpus, terms, postings, searcher
        torch.cuda.empty_cache()
        matrix, corpus, terms, postings = build_rank_state(eng, dev, n_total, D, 0, n_total, args.shadow)
        rq = make_queries(eng, B, D, n_total, stream=rank)
        r_elapsed, _, _, _, _, _ = timed_run(engines, corpus, postings, False, rq, q_terms, q_offs, n_keep, args, world, rehearsal, dev)
        replica = {"queries_per_s": B * args.steps * world / r_elapsed, "ms_per_step": r_elapsed / args.steps * 1e3, "scaling": "weak",
                   "parallelism": f"query-parallel x{world}: corpus replicated, {B} queries per GPU per step, no data-path collective"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = B * args.steps / elapsed * (world if layout == "replica" else 1)
        cross = (3e8 if B <= 2 else 1.5e8) if shadow == "i8" else (6e8 if B <= 2 else 2e8)  # the library's HRS_MODE_AUTO rule
        used_mfma = args.mode == "mfma" or (args.mode == "auto" and n_local >= 65536 and (B > 4 or n_local * D >= cross))
        kern_avg_ms = kern_ms / max(launches, 1)
        if used_mfma:
            # algorithmic: 2*B*N*D per pass of this GPU's shard -- times the share of the tiles that the timed kernel scores
            # itself (the second sampled pass records the rows of its own tiles, about one in 64, and the main pass
            # leaves those out: the library reports the share)
            share = eng.scan_work()
            flops = 2.0 * B * n_local * D * share
            achieved = flops / (kern_avg_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[shadow]
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "kernel": f"k_mfma_scan<main,{shadow}>", "kernel_ms": kern_avg_ms,
                    "main_pass_share_of_tiles": share,
                    "peak_measured": MFMA_MEASURED_TFLOPS[shadow], "frac_of_measured": achieved / MFMA_MEASURED_TFLOPS[shadow],
                    "peak_measured_source": "LDS-fed MFMA loop on random operands (tools/micro/mfma_micro), profiles/r03_mfma_micro.txt"}
        else:
            passes = (B + 3) // 4
            nbytes = float(n_local) * D * 4 * passes  # fp32 corpus streamed once per pass of 4 queries
            achieved = nbytes / (kern_avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": "k_exact_scan", "kernel_ms": kern_avg_ms,
                    "peak_measured": HBM_MEASURED_GBPS, "frac_of_measured": achieved / HBM_MEASURED_GBPS, "peak_measured_source": "copy kernel, MI355X guide"}
        # HBM bytes per launch cannot be counted from inside the run (PMC counters need rocprofv3): what a
        # separate --pmc pass over this same command recorded, with its source, for this exact shape only
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file) and (n_local, D, B) == (10_000_000, 384, 1024):
            try:
                with open(traffic_file) as f:
                    tj = json.load(f)
                if roof["kernel"] in tj:
                    roof["traffic_recorded"] = {"bytes_per_launch": tj[roof["kernel"]], "source": tj.get("source", "profiles/traffic.json"), "how": tj.get("note")}
            except Exception:
                pass
        par = {"single": "single GPU", "shard": f"row-shard x{world}, one all-gather per batch", "replica": f"query-parallel x{world}: corpus replicated, {B} queries per GPU per step, no data-path collective"}[layout]
        line = {
            "metric": f"hybrid top-{n_keep} queries/sec over {n_total // 1_000_000}M x {D} corpus" if (n_total, D, n_keep) != (10_000_000, 384, 10) else "hybrid top-10 queries/sec over 10M x 384 corpus",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if layout == "shard" else "weak",
            "vs_baseline": None,
            "dtype": f"{shadow} MFMA scan + fp64 exact re-score" if used_mfma else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"hybrid {args.space} top-{2 * n_keep} + keyword counts + fusion -> top-{n_keep}; {n_total} x {D} fp32 corpus (unit rows), {VOCAB}-term Zipf postings ({TERMS_PER_DOC}/doc), batch {B}",
                "rows_total": n_total,
                "rows_per_gpu": n_local,
                "dim": D,
                "batch": B,
                "n_results": n_keep,
                "space": args.space,
                "parallelism": par,
                "global_batch": B * (world if layout == "replica" else 1),
                "mode": args.mode,
                "shadow": shadow,
                "streams": len(engines),
            },
            "roofline": roof,
            "fallback_queries_per_batch": stats["fallback_queries"],
            "max_candidates": stats["max_candidates"],
            "checked": checked,
        }
        if world > 1:
            line["rccl"] = {"ranks": world, "backend": "gloo (rehearsal)" if rehearsal else f"{backend} (RCCL)", "collective": "all_gather_into_tensor, one per batch, on its own stream" if layout == "shard" else None,
                            "bytes_per_rank_per_batch": bytes_gathered if layout == "shard" else 0}
        extras = {}
        if replica is not None:
            extras["replica_layout"] = replica
        if world == 1 and not args.no_extras:
            from hybrid_rag_system_amd.dist import ShardedHybrid

            extras["text_queries_end_to_end"] = bench_text_queries(eng, ShardedHybrid(eng, corpus, postings, group=False), n_keep)
            extras["embed"] = bench_embed(eng, cpu_chunks=0 if args.no_cpu_baseline else 2048)
            extras["exact_scan"] = bench_scan(eng, corpus, queries)
            extras["full_corpus_hybrid"] = bench_full_corpus(eng, corpus, terms, queries, q_terms, q_offs, n_keep, engines=engines)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(matrix, terms, queries, q_terms, q_offs, n_keep, n_total, args.cpu_sample_rows, args.cpu_sample_queries)
        else:
            line["cpu_baseline"] = None
        if world == 1 and not args.no_extras:
            corpus.close()
            postings.close()
            del matrix, corpus, terms, postings
            torch.cuda.empty_cache()
            extras.update(bench_other_configs(eng, engines[1] if len(engines) > 1 else Engine(local_rank)))
        if extras:
            line["extras"] = extras
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
