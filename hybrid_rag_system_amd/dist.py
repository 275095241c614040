"""Row-sharding across the GPUs of one node: one process per GPU, torch.distributed (RCCL over xGMI).

The corpus rows (with their term lists) are split contiguously; queries are replicated.  Per query
batch there is exactly one exchange step: an all-gather of every rank's local top-k' keys with the
keyword counts of those candidates riding along (packed into one int64 tensor, so one collective).
Every rank then runs the same deterministic merge and holds the same result.  The payload is tiny
(B * k' * 16 bytes per rank), so the step is latency-bound; nothing is reduced and no ring is built.
The reference has no counterpart: it is a single process (scripts/chromadb_helper.py:591-684).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world, rank):
    """Contiguous row range [lo, hi) of `rank` out of `world` shards."""
    return n_rows * rank // world, n_rows * (rank + 1) // world


def pack_candidates(keys, counts):
    """keys int64 [B,c] (uint64 bits) + counts [B,c] (int32 keyword counts, or float64 keyword scores, which travel as
    their bits) -> int64 [B,c,2]."""
    second = counts.view(torch.int64) if counts.dtype == torch.float64 else counts.to(torch.int64)
    return torch.stack((keys, second), dim=-1).contiguous()


def allgather_candidates(keys, counts, group=None):
    """All-gather every rank's (keys, counts) -> keys int64 [G,B,c], counts [G,B,c] (int32, or float64 as given)."""
    world = dist.get_world_size(group)
    payload = pack_candidates(keys, counts)
    # concatenated along dim 0 (the form both RCCL and gloo accept), viewed as [G, B, c, 2]
    if payload.is_cuda and dist.get_backend(group) == "gloo":
        # gloo (rehearsals and CPU tests only) moves device tensors at a crawl: stage through the host
        host = payload.cpu()
        flat_host = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(flat_host, host, group=group)
        flat = flat_host.to(payload.device)
    else:
        flat = torch.empty((world * payload.shape[0],) + tuple(payload.shape[1:]), dtype=payload.dtype, device=payload.device)
        dist.all_gather_into_tensor(flat, payload, group=group)
    out = flat.view((world,) + tuple(payload.shape))
    second = out[..., 1].contiguous()
    return out[..., 0].contiguous(), (second.view(torch.float64) if counts.dtype == torch.float64 else second.to(torch.int32))


def exchange_classes(engine, group=None):
    """-> the `exchange` step of Engine.hybrid_full_corpus on a row-sharded corpus: one all-gather of every shard's
    per-class candidates (keys int64 [B, levels, n], local rows already carry the shard's row offset), then the
    per-class merge on every rank."""

    def exchange(keys):
        world = dist.get_world_size(group)
        staged = keys.is_cuda and dist.get_backend(group) == "gloo"  # see allgather_candidates
        src = (keys.cpu() if staged else keys).contiguous()
        flat = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(flat, src, group=group)
        gathered = flat.view((world,) + tuple(src.shape))
        return engine.merge_classes(gathered.to(keys.device) if staged else gathered)

    return exchange


def embed_data_parallel(embed_fn, texts, group=None):
    """Indexing is data-parallel: rank r embeds the contiguous slice of `texts` that shard_bounds gives it
    (the rows it will own) with `embed_fn(list[str]) -> fp32 [n, D]`, and nothing is exchanged.
    -> (lo, hi, rows[lo:hi]).  `gather_rows` assembles the full matrix where a caller wants a replica."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(texts), world, rank)
    return lo, hi, embed_fn(list(texts[lo:hi]))


def gather_rows(local_rows, n_total, group=None):
    """All-gather contiguous row shards (sizes from shard_bounds, so they differ by at most one row)
    -> [n_total, D] on every rank: one collective, shards padded to the longest."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_rows
    world = dist.get_world_size(group)
    longest = max(hi - lo for lo, hi in (shard_bounds(n_total, world, r) for r in range(world)))
    pad = torch.zeros((longest, local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    staged = pad.is_cuda and dist.get_backend(group) == "gloo"  # see allgather_candidates
    src = pad.cpu() if staged else pad
    flat = torch.empty((world * longest, local_rows.shape[1]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(flat, src, group=group)
    parts = [flat[r * longest : r * longest + (shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0])] for r in range(world)]
    out = torch.cat(parts)
    return out.to(local_rows.device) if staged else out


def bm25_corpus_stats(local_doc_freq, local_total_len, local_rows, group=None, device=None):
    """The BM25 statistics of a row-sharded corpus, identical on every rank: one all-reduce (sum) of the shard's
    document frequencies int64 [V], its summed document lengths and its row count, then
    idf = ln(1 + (N - df + 0.5) / (df + 0.5)) in numpy float64 on the host (integers in, the same libm on every rank).
    -> dict(idf=fp64 [V] on `device`, avg_len=float): the `bm25` argument of ShardedHybrid.step / Engine.hybrid_topk."""
    import numpy as np

    df = torch.as_tensor(local_doc_freq, dtype=torch.int64).cpu().reshape(-1)
    packed = torch.cat((df, torch.tensor([int(local_total_len), int(local_rows)], dtype=torch.int64)))
    if dist.is_initialized() and group is not False and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "gloo":
            dist.all_reduce(packed, group=group)
        else:  # RCCL reduces device tensors
            dev = packed.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(dev, group=group)
            packed = dev.cpu()
    tot = packed.numpy()
    n_rows, total_len = int(tot[-1]), int(tot[-2])
    dfa = tot[:-2].astype(np.float64)
    idf = np.log(1.0 + (float(n_rows) - dfa + 0.5) / (dfa + 0.5))
    out = torch.from_numpy(idf)
    return {"idf": out.to(device) if device is not None else out, "avg_len": float(total_len) / float(max(n_rows, 1))}


def _is_cuda(engine):
    return getattr(getattr(engine, "device", None), "type", "cpu") == "cuda"


class ShardedHybrid:
    """Hybrid top-n over a row-sharded corpus; every rank returns the same merged result.
    group=False: this rank holds the whole corpus (single GPU, or one replica of several) and no
    exchange takes place even when a process group exists.

    exchange_stream (a torch.cuda.Stream, optional): the exchange step -- all-gather of the per-shard
    candidates, merge, fusion -- is enqueued there instead of on the stream of the scan, behind an event, so the
    scan's stream is free for the next batch while the (latency-bound) collective is in flight.  The returned
    tensors are then valid on that stream; `done_event` marks their completion."""

    def __init__(self, engine, corpus, postings, group=None, exchange_stream=None):
        self.engine, self.corpus, self.postings, self.group = engine, corpus, postings, group
        self.exchange_stream = exchange_stream
        self.done_event = None
        self._open = None
        self.bytes_gathered = 0  # payload this rank contributed to the last exchange

    def exchanges(self):
        return self.group is not False and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def step(self, queries, q_terms, q_term_offs, n, boost=0.3, space="l2", mode="auto", bm25=None):
        self.begin(queries, q_terms, q_term_offs, n, boost, space, mode, bm25)
        return self.finish()

    def begin(self, queries, q_terms, q_term_offs, n, boost=0.3, space="l2", mode="auto", bm25=None):
        """The scan of a step (hrs_score_topk_begin); `finish` on the same stream does the rest.
        bm25 = dict(idf=fp64 [V], avg_len=float[, k1, b]) as in Engine.hybrid_topk, with the statistics of the WHOLE
        corpus (document frequencies and lengths summed over the shards when the collection was built): every shard
        then weighs its own candidates exactly as an unsharded engine would, and the scores ride along in the
        all-gather where the counts do otherwise (same payload size)."""
        # local top-2n of this shard is a superset of the shard's members of the global top-2n
        self.engine.score_topk_begin(self.corpus, queries, 2 * n, space=space, mode=mode)
        self._open = (q_terms, q_term_offs, n, boost, bm25)

    def finish(self):
        e = self.engine
        q_terms, q_term_offs, n, boost, bm25 = self._open
        self._open = None
        cand_dist, rows, keys = e.score_topk_finish(want_keys=True)
        if bm25 is not None:
            counts = e.bm25_scores(self.postings, bm25["idf"], bm25["avg_len"], q_terms, q_term_offs, rows, bm25.get("k1", 1.2), bm25.get("b", 0.75))
        else:
            counts = e.keyword_counts(self.postings, q_terms, q_term_offs, rows)
        local = (cand_dist, rows)  # the candidates unpacked, as the scoring call wrote them
        cuda = _is_cuda(e)
        xs = self.exchange_stream if (cuda and self.exchanges()) else None
        if xs is not None:
            ready = torch.cuda.Event()
            ready.record()  # on the scan's stream: candidates and their counts are complete
            for t in (keys, counts, q_term_offs):
                t.record_stream(xs)  # allocated on the scan's stream, read on the exchange stream
            with torch.cuda.stream(xs):
                xs.wait_event(ready)
                out = self._exchange_and_fuse(keys, counts, q_term_offs, n, boost, local)
                self.done_event = torch.cuda.Event()
                self.done_event.record()
            return out
        out = self._exchange_and_fuse(keys, counts, q_term_offs, n, boost, local)
        if cuda:
            self.done_event = torch.cuda.Event()
            self.done_event.record()
        return out

    def _exchange_and_fuse(self, keys, counts, q_term_offs, n, boost, local):
        e = self.engine
        if self.exchanges():
            self.bytes_gathered = keys.numel() * 16  # (key, count) as two int64 per candidate
            gk, gc = allgather_candidates(keys, counts, self.group)
            keys, counts = e.merge_topk(gk, gc)
            cand_dist, cand_rows = e.unpack_keys(keys)
        else:
            cand_dist, cand_rows = local
        if counts.dtype == torch.float64:  # BM25 keyword scores
            combined, perm = e.fuse_sort_scores(cand_dist, counts, n, boost)
            return {"cand_dist": cand_dist, "cand_rows": cand_rows, "cand_kw": counts, "combined": combined, "perm": perm}
        combined, perm = e.fuse_sort(cand_dist, counts, q_term_offs, n, boost)
        return {"cand_dist": cand_dist, "cand_rows": cand_rows, "cand_counts": counts, "combined": combined, "perm": perm}


class _NoStream:
    """Stands where a HIP stream would on an engine without one (CPU rehearsals of the pipeline's order of events)."""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class HybridPipeline:
    """Consecutive query batches on alternating HIP streams, one hrs_ctx (own workspaces) per stream.

    A batch is a chain of kernels with very different appetites: the matrix-core scan fills the MFMA
    pipes, the candidate sort / gather / exact re-score that follows is bound by the latency of
    scattered row reads.  The scans take turns (hrs_ctx_set_peer); `submit` begins batch i + 1 on the
    next stream and only then enqueues the tail of batch i, which runs right behind its own main pass, beside
    the sampled passes of batch i + 1 (the two share the chip at workgroup granularity; queued under the next
    main pass instead, the tail only ran when that pass retired).  On a row-sharded corpus the
    exchange step of batch i (all-gather, merge, fusion) goes to a third stream of its own, so it too is in
    flight under the scan of batch i + 1.  Every batch runs every kernel of ShardedHybrid.step and results
    are identical.

    Stream contract: `submit` orders the work behind whatever the caller's current stream has enqueued (the
    inputs may have been produced there) and tells the allocator that they are in use on the pipeline's
    streams, so the caller may drop them right after the call.  It returns the tensors of the PREVIOUS batch
    (None for the first) after making the caller's current stream wait for them: they can be consumed there
    without a device-wide synchronisation.  `drain()` finishes the last batch the same way.
    """

    def __init__(self, engines, corpus, postings, group=None):
        self.cuda = _is_cuda(engines[0])
        mk = (lambda e: torch.cuda.Stream(device=e.device)) if self.cuda else (lambda e: None)
        self.lanes = []
        for e in engines:
            searcher = ShardedHybrid(e, corpus, postings, group)
            if self.cuda and searcher.exchanges():
                searcher.exchange_stream = mk(e)
            self.lanes.append((searcher, mk(e)))
        # the scans themselves take turns: each engine's scan waits for the previous engine's
        for j, e in enumerate(engines):
            if len(engines) > 1:
                e.set_peer(engines[j - 1])
        self.n = 0
        self._open = None  # lane with a begun, unfinished batch

    def _on(self, stream):
        return torch.cuda.stream(stream) if self.cuda else _NoStream()

    def _finish_open(self):
        """Enqueue the tail of the open batch; the caller's stream then waits for its results."""
        if self._open is None:
            return None
        searcher, stream = self._open
        self._open = None
        with self._on(stream):
            out = searcher.finish()
        if self.cuda:
            cur = torch.cuda.current_stream()
            cur.wait_event(searcher.done_event)
            for t in out.values():
                t.record_stream(cur)  # allocated on a pipeline stream, consumed on the caller's
        return out

    def submit(self, queries, q_terms, q_term_offs, n, **kw):
        searcher, stream = self.lanes[self.n % len(self.lanes)]
        self.n += 1
        if len(self.lanes) == 1:
            prev = self._finish_open()
        if self.cuda:
            stream.wait_stream(torch.cuda.current_stream())  # the inputs were produced on the caller's stream
            for t in (queries, q_terms, q_term_offs):
                t.record_stream(stream)
        with self._on(stream):
            searcher.begin(queries, q_terms, q_term_offs, n, **kw)
        if len(self.lanes) > 1:
            prev = self._finish_open()  # after the next scan has been enqueued
        self._open = (searcher, stream)
        return prev

    def drain(self):
        last = self._finish_open()
        if self.cuda:
            for searcher, stream in self.lanes:
                torch.cuda.current_stream().wait_stream(stream)
                if searcher.exchange_stream is not None:
                    torch.cuda.current_stream().wait_stream(searcher.exchange_stream)
        return last


class TopKPipeline:
    """Plain top-k scoring (collection.query without the keyword stage) with consecutive batches on alternating
    streams / contexts, as HybridPipeline does for the hybrid step: batch i + 1's scan is enqueued before batch i's
    tail, which then runs under it.  `submit` returns the (dist, rows, keys) of the previous batch, consumable on the
    caller's stream; `drain` the last."""

    def __init__(self, engines, corpus):
        self.corpus = corpus
        self.lanes = [(e, torch.cuda.Stream(device=e.device)) for e in engines]
        for j, e in enumerate(engines):
            if len(engines) > 1:
                e.set_peer(engines[j - 1])
        self.n = 0
        self._open = None

    def _finish_open(self):
        if self._open is None:
            return None
        eng, stream = self._open
        self._open = None
        with torch.cuda.stream(stream):
            out = eng.score_topk_finish(want_keys=True)
            done = torch.cuda.Event()
            done.record()
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        for t in out:
            t.record_stream(cur)
        return out

    def submit(self, queries, k, space="l2", mask=None, mode="auto"):
        eng, stream = self.lanes[self.n % len(self.lanes)]
        self.n += 1
        prev = self._finish_open() if len(self.lanes) == 1 else None
        stream.wait_stream(torch.cuda.current_stream())
        queries.record_stream(stream)
        with torch.cuda.stream(stream):
            eng.score_topk_begin(self.corpus, queries, k, space=space, mask=mask, mode=mode)
        if len(self.lanes) > 1:
            prev = self._finish_open()
        self._open = (eng, stream)
        return prev

    def drain(self):
        last = self._finish_open()
        for _, stream in self.lanes:
            torch.cuda.current_stream().wait_stream(stream)
        return last
