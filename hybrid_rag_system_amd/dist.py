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
    """keys int64 [B,c] (uint64 bits) + counts int32 [B,c] -> int64 [B,c,2]."""
    return torch.stack((keys, counts.to(torch.int64)), dim=-1).contiguous()


def allgather_candidates(keys, counts, group=None):
    """All-gather every rank's (keys, counts) -> keys int64 [G,B,c], counts int32 [G,B,c]."""
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
    return out[..., 0].contiguous(), out[..., 1].to(torch.int32).contiguous()


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


class ShardedHybrid:
    """Hybrid top-n over a row-sharded corpus; every rank returns the same merged result.
    group=False: this rank holds the whole corpus (single GPU, or one replica of several) and no
    exchange takes place even when a process group exists."""

    def __init__(self, engine, corpus, postings, group=None):
        self.engine, self.corpus, self.postings, self.group = engine, corpus, postings, group

    def step(self, queries, q_terms, q_term_offs, n, boost=0.3, space="l2", mode="auto"):
        self.begin(queries, q_terms, q_term_offs, n, boost, space, mode)
        return self.finish()

    def begin(self, queries, q_terms, q_term_offs, n, boost=0.3, space="l2", mode="auto"):
        """The scan of a step (hrs_score_topk_begin); `finish` on the same stream does the rest."""
        # local top-2n of this shard is a superset of the shard's members of the global top-2n
        self.engine.score_topk_begin(self.corpus, queries, 2 * n, space=space, mode=mode)
        self._open = (q_terms, q_term_offs, n, boost)

    def finish(self):
        e = self.engine
        q_terms, q_term_offs, n, boost = self._open
        self._open = None
        _, rows, keys = e.score_topk_finish(want_keys=True)
        counts = e.keyword_counts(self.postings, q_terms, q_term_offs, rows)
        if self.group is not False and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            gk, gc = allgather_candidates(keys, counts, self.group)
            keys, counts = e.merge_topk(gk, gc)
        cand_dist, cand_rows = e.unpack_keys(keys)
        combined, perm = e.fuse_sort(cand_dist, counts, q_term_offs, n, boost)
        return {"cand_dist": cand_dist, "cand_rows": cand_rows, "cand_counts": counts, "combined": combined, "perm": perm}


class HybridPipeline:
    """Consecutive query batches on alternating HIP streams, one hrs_ctx (own workspaces) per stream.

    A batch is a chain of kernels with very different appetites: the matrix-core scan fills the MFMA
    pipes, the candidate sort / gather / exact re-score that follows is bound by the latency of
    scattered row reads.  The scans take turns (hrs_ctx_set_peer); `submit` begins batch i + 1 on the
    next stream and only then enqueues the tail of batch i, which the library holds back until batch
    i + 1 has its thresholds: the tail then runs under the next main pass (both fit a CU's LDS and
    registers together) instead of beside the sampled passes that precede it.  Every batch runs every
    kernel of ShardedHybrid.step and results are identical.  `submit` returns the tensors of the
    PREVIOUS batch (None for the first); `drain()` finishes the last one, waits for the streams on the
    caller's stream and returns its tensors.
    """

    def __init__(self, engines, corpus, postings, group=None):
        self.lanes = [(ShardedHybrid(e, corpus, postings, group), torch.cuda.Stream(device=e.device)) for e in engines]
        # the scans themselves take turns: each engine's scan waits for the previous engine's
        for j, e in enumerate(engines):
            if len(engines) > 1:
                e.set_peer(engines[j - 1])
        self.n = 0
        self._open = None  # lane with a begun, unfinished batch

    def _finish_open(self):
        if self._open is None:
            return None
        searcher, stream = self._open
        self._open = None
        with torch.cuda.stream(stream):
            return searcher.finish()

    def submit(self, queries, q_terms, q_term_offs, n, **kw):
        searcher, stream = self.lanes[self.n % len(self.lanes)]
        self.n += 1
        if len(self.lanes) == 1:
            prev = self._finish_open()
        stream.wait_stream(torch.cuda.current_stream())  # the inputs were produced on the caller's stream
        with torch.cuda.stream(stream):
            searcher.begin(queries, q_terms, q_term_offs, n, **kw)
        if len(self.lanes) > 1:
            prev = self._finish_open()  # after the next scan has been enqueued
        self._open = (searcher, stream)
        return prev

    def drain(self):
        last = self._finish_open()
        for _, stream in self.lanes:
            torch.cuda.current_stream().wait_stream(stream)
        return last
