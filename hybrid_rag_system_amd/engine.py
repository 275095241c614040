"""Host-side handles over the C ABI: one Engine per GPU, torch tensors as device memory.

torch is plumbing here (allocation, streams, torch.distributed); every computation on the retrieval
path is a HIP kernel of libhrs.so.  Nothing in this module computes on the CPU and no query falls
back to torch ops: a missing library or GPU raises.  (The one torch computation here is an index
build, InvertedPostings.from_forward, which sorts (term, row) pairs once per collection.)
"""
import ctypes as C

import torch

from . import _native as N


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Corpus:
    """The embedding matrix of one shard, registered with the engine (keeps `matrix` alive)."""

    def __init__(self, engine, matrix, row_offset=0, shadow="auto", n=None):
        if not (matrix.is_cuda and matrix.dtype == torch.float32 and matrix.dim() == 2 and matrix.is_contiguous()):
            raise ValueError("corpus must be a contiguous fp32 [N, D] CUDA tensor")
        self.engine = engine
        self.matrix = matrix
        self.n, self.dim = matrix.shape
        if n is not None:  # the first n rows of a buffer that has room for more (see append)
            if not 0 <= n <= matrix.shape[0]:
                raise ValueError(f"n={n} outside the matrix ({matrix.shape[0]} rows)")
            self.n = int(n)
        self.row_offset = int(row_offset)
        h = C.c_void_p()
        N.check(
            N.lib().hrs_corpus_register(engine._ctx, _ptr(matrix) if self.n else None, self.n, self.dim, self.row_offset, _stream(), C.byref(h)),
            "hrs_corpus_register",
        )
        self._h = h
        if shadow != "auto":
            self.set_shadow(shadow)

    def append(self, matrix, n=None):
        """The collection grew: `matrix` (fp32 [>= n, D], contiguous; rows [0, self.n) as registered, possibly at a
        new address) now holds n rows.  Only the new rows are quantised (hrs_corpus_append)."""
        n = matrix.shape[0] if n is None else int(n)
        if not (matrix.is_cuda and matrix.dtype == torch.float32 and matrix.dim() == 2 and matrix.is_contiguous() and matrix.shape[1] == self.dim):
            raise ValueError("corpus must stay a contiguous fp32 [N, D] CUDA tensor of the registered depth")
        if n > matrix.shape[0] or n < self.n:
            raise ValueError(f"append to {n} rows: the matrix holds {matrix.shape[0]}, the corpus {self.n}")
        N.check(N.lib().hrs_corpus_append(self.engine._ctx, self._h, _ptr(matrix) if n else None, n, _stream()), "hrs_corpus_append")
        self.matrix, self.n = matrix, n

    def set_shadow(self, kind):
        """Rebuild the scan shadow: "auto" (int8), "i8", "bf16" or "none" (exact scan only)."""
        N.check(N.lib().hrs_corpus_shadow(self.engine._ctx, self._h, N.SHADOWS[kind], _stream()), "hrs_corpus_shadow")

    def norm_stats(self):
        mn, mx = C.c_double(), C.c_double()
        N.check(N.lib().hrs_corpus_stats(self.engine._ctx, self._h, _stream(), C.byref(mn), C.byref(mx)), "hrs_corpus_stats")
        return mn.value, mx.value

    def close(self):
        if self._h is not None:
            N.lib().hrs_corpus_release(self.engine._ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Postings:
    """Forward CSR term lists of one shard (keeps both tensors alive)."""

    def __init__(self, engine, doc_offs, term_ids, row_offset=0):
        if not (doc_offs.is_cuda and doc_offs.dtype == torch.int64 and doc_offs.is_contiguous()):
            raise ValueError("doc_offs must be a contiguous int64 CUDA tensor")
        if not (term_ids.is_cuda and term_ids.dtype == torch.int32 and term_ids.is_contiguous()):
            raise ValueError("term_ids must be a contiguous int32 CUDA tensor")
        self.engine, self.doc_offs, self.term_ids = engine, doc_offs, term_ids
        self.n = doc_offs.numel() - 1
        self.row_offset = int(row_offset)
        h = C.c_void_p()
        N.check(N.lib().hrs_postings_register(engine._ctx, _ptr(doc_offs), _ptr(term_ids), self.n, self.row_offset, C.byref(h)), "hrs_postings_register")
        self._h = h

    def close(self):
        if self._h is not None:
            N.lib().hrs_postings_release(self.engine._ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class InvertedPostings:
    """Inverted CSR of one shard: rows of term t are doc_ids[term_offs[t] : term_offs[t + 1]] (keeps both tensors alive)."""

    def __init__(self, engine, term_offs, doc_ids, n_rows):
        if not (term_offs.is_cuda and term_offs.dtype == torch.int64 and term_offs.is_contiguous()):
            raise ValueError("term_offs must be a contiguous int64 CUDA tensor")
        if not (doc_ids.is_cuda and doc_ids.dtype == torch.int32 and doc_ids.is_contiguous()):
            raise ValueError("doc_ids must be a contiguous int32 CUDA tensor")
        self.engine, self.term_offs, self.doc_ids = engine, term_offs, doc_ids
        self.vocab, self.n = term_offs.numel() - 1, int(n_rows)
        h = C.c_void_p()
        N.check(N.lib().hrs_postings_register_inverted(engine._ctx, _ptr(term_offs), _ptr(doc_ids), self.vocab, self.n, C.byref(h)), "hrs_postings_register_inverted")
        self._h = h

    @classmethod
    def from_forward(cls, engine, doc_offs, term_ids, vocab):
        """Build the inverted lists from forward CSR term lists (torch sorting as plumbing: an index build, not a query path)."""
        n = doc_offs.numel() - 1
        rows = torch.repeat_interleave(torch.arange(n, device=term_ids.device, dtype=torch.int32), (doc_offs[1:] - doc_offs[:-1]))
        key = term_ids.to(torch.int64) * n + rows.to(torch.int64)
        key = torch.unique(key)  # sorted; a term listed twice in a row counts once
        terms, docs = key // n, (key % n).to(torch.int32)
        offs = torch.zeros(vocab + 1, dtype=torch.int64, device=term_ids.device)
        offs[1:] = torch.cumsum(torch.bincount(terms, minlength=vocab), 0)
        return cls(engine, offs.contiguous(), docs.contiguous(), n)

    def close(self):
        if self._h is not None:
            N.lib().hrs_postings_release_inverted(self.engine._ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Encoder:
    """MiniLM-class encoder resident on the GPU (owns its weight copies)."""

    def __init__(self, engine, weights, cfg):
        from .weights import tensor_specs

        self.engine, self.cfg = engine, cfg
        specs = tensor_specs(cfg)
        dev = [torch.as_tensor(weights[name], dtype=torch.float32).contiguous().to(engine.device) for name, _ in specs]
        for t, (name, shape) in zip(dev, specs):
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {shape}")
        ptrs = (C.c_void_p * len(dev))(*[t.data_ptr() for t in dev])
        c = N.EncoderConfigC(cfg.vocab, cfg.hidden, cfg.layers, cfg.heads, cfg.inter, cfg.max_pos, cfg.eps)
        h = C.c_void_p()
        N.check(N.lib().hrs_encoder_load(engine._ctx, C.byref(c), ptrs, _stream(), C.byref(h)), "hrs_encoder_load")
        torch.cuda.current_stream().synchronize()  # the fp32 staging tensors may go now
        self._h = h

    def encode_ids(self, ids, lens):
        """ids int32 [B,S] (S a multiple of 32, <= 256), lens int32 [B] -> fp32 [B, hidden] unit rows."""
        if not (ids.is_cuda and ids.dtype == torch.int32 and ids.dim() == 2 and ids.is_contiguous()):
            raise ValueError("ids must be a contiguous int32 [B, S] CUDA tensor")
        if not (lens.is_cuda and lens.dtype == torch.int32 and lens.numel() == ids.shape[0]):
            raise ValueError("lens must be an int32 [B] CUDA tensor")
        B, S = ids.shape
        out = torch.empty((B, self.cfg.hidden), dtype=torch.float32, device=self.engine.device)
        N.check(N.lib().hrs_encode(self.engine._ctx, self._h, _ptr(ids), _ptr(lens), B, S, _ptr(out), _stream()), "hrs_encode")
        return out

    def close(self):
        if self._h is not None:
            N.lib().hrs_encoder_release(self.engine._ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One hrs_ctx on one MI355X."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise N.HrsError("no GPU visible: hybrid_rag_system_amd runs on MI355X (gfx950) only and has no CPU path")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        N.check(N.lib().hrs_ctx_create(device, C.byref(h)), "hrs_ctx_create")
        self._ctx = h

    def close(self):
        if self._ctx is not None:
            N.lib().hrs_ctx_destroy(self._ctx)
            self._ctx = None

    # ---- registration -------------------------------------------------------------------------
    def set_peer(self, other):
        """Scans launched through this engine wait for the last scan launched through `other` (see HybridPipeline)."""
        N.check(N.lib().hrs_ctx_set_peer(self._ctx, other._ctx if other is not None else None), "hrs_ctx_set_peer")

    def corpus(self, matrix, row_offset=0, shadow="auto", n=None):
        return Corpus(self, matrix, row_offset, shadow, n)

    def postings(self, doc_offs, term_ids, row_offset=0):
        return Postings(self, doc_offs, term_ids, row_offset)

    def inverted_postings(self, term_offs, doc_ids, n_rows):
        return InvertedPostings(self, term_offs, doc_ids, n_rows)

    def encoder(self, weights, cfg):
        return Encoder(self, weights, cfg)

    # ---- scoring -------------------------------------------------------------------------------
    def score_topk(self, corpus, queries, k, space="l2", mask=None, mode="auto", want_keys=False):
        """-> (dist fp32 [B,k], rows int64 [B,k][, keys uint64-as-int64 [B,k]]) on the GPU."""
        if not (queries.is_cuda and queries.dtype == torch.float32 and queries.dim() == 2 and queries.is_contiguous()):
            raise ValueError("queries must be a contiguous fp32 [B, D] CUDA tensor")
        if queries.shape[1] != corpus.dim:
            raise ValueError(f"query dim {queries.shape[1]} != corpus dim {corpus.dim}")
        B = queries.shape[0]
        dist = torch.empty((B, k), dtype=torch.float32, device=self.device)
        rows = torch.empty((B, k), dtype=torch.int64, device=self.device)
        keys = torch.empty((B, k), dtype=torch.int64, device=self.device)
        m = {"auto": N.MODE_AUTO, "exact": N.MODE_EXACT, "mfma": N.MODE_MFMA}[mode]
        N.check(
            N.lib().hrs_score_topk(self._ctx, corpus._h, _ptr(queries), B, k, N.SPACES[space], _ptr(mask), m, _ptr(keys), _ptr(dist), _ptr(rows), _stream()),
            "hrs_score_topk",
        )
        return (dist, rows, keys) if want_keys else (dist, rows)

    def score_topk_after(self, corpus, queries, k, after_keys, space="l2", mask=None):
        """score_topk resumed behind `after_keys` (int64 [B], uint64 bits: the last key each query has received)
        -> (dist, rows, keys)."""
        B = queries.shape[0]
        dist = torch.empty((B, k), dtype=torch.float32, device=self.device)
        rows = torch.empty((B, k), dtype=torch.int64, device=self.device)
        keys = torch.empty((B, k), dtype=torch.int64, device=self.device)
        N.check(
            N.lib().hrs_score_topk_after(self._ctx, corpus._h, _ptr(queries), B, k, N.SPACES[space], _ptr(mask), _ptr(after_keys), _ptr(keys), _ptr(dist), _ptr(rows), _stream()),
            "hrs_score_topk_after",
        )
        return dist, rows, keys

    def score_topk_begin(self, corpus, queries, k, space="l2", mask=None, mode="auto"):
        """First half of score_topk (the scan); `queries` and `mask` must stay alive until score_topk_finish."""
        if not (queries.is_cuda and queries.dtype == torch.float32 and queries.dim() == 2 and queries.is_contiguous()):
            raise ValueError("queries must be a contiguous fp32 [B, D] CUDA tensor")
        if queries.shape[1] != corpus.dim:
            raise ValueError(f"query dim {queries.shape[1]} != corpus dim {corpus.dim}")
        m = {"auto": N.MODE_AUTO, "exact": N.MODE_EXACT, "mfma": N.MODE_MFMA}[mode]
        N.check(
            N.lib().hrs_score_topk_begin(self._ctx, corpus._h, _ptr(queries), queries.shape[0], k, N.SPACES[space], _ptr(mask), m, _stream()),
            "hrs_score_topk_begin",
        )
        self._open_call = (queries, mask, queries.shape[0], k)

    def score_topk_finish(self, want_keys=False):
        """Second half (sort, exact re-score, proof, fallback) on the same stream -> what score_topk returns."""
        if getattr(self, "_open_call", None) is None:
            raise ValueError("score_topk_finish without score_topk_begin")
        _, _, B, k = self._open_call
        dist = torch.empty((B, k), dtype=torch.float32, device=self.device)
        rows = torch.empty((B, k), dtype=torch.int64, device=self.device)
        keys = torch.empty((B, k), dtype=torch.int64, device=self.device)
        N.check(N.lib().hrs_score_topk_finish(self._ctx, _ptr(keys), _ptr(dist), _ptr(rows), _stream()), "hrs_score_topk_finish")
        self._open_call = None
        return (dist, rows, keys) if want_keys else (dist, rows)

    def score_stats(self):
        out = (C.c_int64 * 3)()
        N.check(N.lib().hrs_score_stats(self._ctx, _stream(), out), "hrs_score_stats")
        return {"fallback_queries": out[0], "max_candidates": out[1], "overflow_queries": out[2]}

    def scan_work(self):
        """-> share of the corpus' tiles that the last matrix-core call's main pass scored itself (see hrs_scan_work)."""
        share = C.c_double()
        N.check(N.lib().hrs_scan_work(self._ctx, _stream(), C.byref(share)), "hrs_scan_work")
        return share.value

    def profile(self, on=True):
        N.check(N.lib().hrs_profile_enable(self._ctx, int(on)), "hrs_profile_enable")

    def profile_read(self):
        """-> (summed ms of the dominant scan kernel since the last read, number of launches)."""
        ms, n = C.c_double(), C.c_int64()
        N.check(N.lib().hrs_profile_read(self._ctx, C.byref(ms), C.byref(n)), "hrs_profile_read")
        return ms.value, n.value

    def row_mask_update(self, mask, codes=None, values=None, negate=False, other=None, combine="set"):
        """mask uint8 [N] <- combine(mask, codes in values (xor negate))  or  combine(mask, other); see hrs_row_mask_update."""
        m = values.numel() if values is not None else 0
        N.check(
            N.lib().hrs_row_mask_update(self._ctx, mask.numel(), _ptr(codes), _ptr(values), m, int(bool(negate)), _ptr(other), {"set": 0, "and": 1, "or": 2}[combine], _ptr(mask), _stream()),
            "hrs_row_mask_update",
        )
        return mask

    def keyword_counts(self, postings, q_terms, q_term_offs, cand_rows):
        B, c = cand_rows.shape
        counts = torch.empty((B, c), dtype=torch.int32, device=self.device)
        N.check(
            N.lib().hrs_keyword_counts(self._ctx, postings._h, _ptr(q_terms), _ptr(q_term_offs), _ptr(cand_rows), B, c, _ptr(counts), _stream()),
            "hrs_keyword_counts",
        )
        return counts

    def fuse_sort(self, dist, counts, q_term_offs, n, boost=0.3):
        B, c = dist.shape
        combined = torch.empty((B, n), dtype=torch.float64, device=self.device)
        perm = torch.empty((B, n), dtype=torch.int32, device=self.device)
        N.check(
            N.lib().hrs_fuse_sort(self._ctx, _ptr(dist), _ptr(counts), _ptr(q_term_offs), B, c, float(boost), n, _ptr(combined), _ptr(perm), _stream()),
            "hrs_fuse_sort",
        )
        return combined, perm

    def bm25_scores(self, postings, idf, avg_len, q_terms, q_term_offs, cand_rows, k1=1.2, b=0.75):
        """BM25 keyword score of every candidate, in [0, 1) (an extension; see hrs_bm25_scores): idf fp64 [V] on the GPU."""
        if not (idf.is_cuda and idf.dtype == torch.float64 and idf.is_contiguous()):
            raise ValueError("idf must be a contiguous float64 CUDA tensor")
        B, c = cand_rows.shape
        scores = torch.empty((B, c), dtype=torch.float64, device=self.device)
        N.check(
            N.lib().hrs_bm25_scores(self._ctx, postings._h, _ptr(idf), idf.numel(), float(avg_len), float(k1), float(b), _ptr(q_terms), _ptr(q_term_offs), _ptr(cand_rows), B, c,
                                    _ptr(scores), _stream()),
            "hrs_bm25_scores",
        )
        return scores

    def fuse_sort_scores(self, dist, kw_scores, n, boost=0.3):
        B, c = dist.shape
        combined = torch.empty((B, n), dtype=torch.float64, device=self.device)
        perm = torch.empty((B, n), dtype=torch.int32, device=self.device)
        N.check(N.lib().hrs_fuse_sort_scores(self._ctx, _ptr(dist), _ptr(kw_scores), B, c, float(boost), n, _ptr(combined), _ptr(perm), _stream()), "hrs_fuse_sort_scores")
        return combined, perm

    def match_counts(self, inverted, q_terms, q_term_offs, B):
        """-> uint8 [B, pitch] (pitch = rows rounded up to 4): per query, how many of its terms every row holds."""
        pitch = (inverted.n + 3) // 4 * 4
        counts = torch.empty((B, pitch), dtype=torch.uint8, device=self.device)
        N.check(N.lib().hrs_match_counts(self._ctx, inverted._h, _ptr(q_terms), _ptr(q_term_offs), B, pitch, _ptr(counts), _stream()), "hrs_match_counts")
        return counts

    def score_topk_levels(self, corpus, queries, k, levels, level_of_row, space="l2"):
        """level_of_row uint8 [B, pitch] -> keys int64 (uint64 bits) [B, levels, k]: exact top-k within every level."""
        B = queries.shape[0]
        keys = torch.empty((B, levels, k), dtype=torch.int64, device=self.device)
        N.check(
            N.lib().hrs_score_topk_levels(self._ctx, corpus._h, _ptr(queries), B, k, N.SPACES[space], _ptr(level_of_row), level_of_row.shape[1], levels, _ptr(keys), _stream()),
            "hrs_score_topk_levels",
        )
        return keys

    def full_corpus_local(self, corpus, inverted, queries, q_terms, q_term_offs, n, levels, space="l2"):
        """First half of the full-corpus hybrid mode on one shard: match counts of every local row, then the n nearest
        rows of every match-count class -> keys int64 (uint64 bits) [B, levels, n]."""
        counts = self.match_counts(inverted, q_terms, q_term_offs, queries.shape[0])
        return self.score_topk_levels(corpus, queries, n, levels, counts, space)

    def full_corpus_finish(self, keys, q_term_offs, n, boost=0.3):
        """Second half: keys [B, levels, n] (of one shard, or merged per class over all shards) -> the n best rows under
        the fused score.  The class index IS the keyword match count."""
        B, levels, k = keys.shape
        out = {"rows": torch.empty((B, n), dtype=torch.int64, device=self.device), "dist": torch.empty((B, n), dtype=torch.float32, device=self.device),
               "counts": torch.empty((B, n), dtype=torch.int32, device=self.device), "combined": torch.empty((B, n), dtype=torch.float64, device=self.device)}
        N.check(
            N.lib().hrs_classes_finish(self._ctx, _ptr(keys), _ptr(q_term_offs), B, levels, k, float(boost), n, _ptr(out["rows"]), _ptr(out["dist"]),
                                       _ptr(out["counts"]), _ptr(out["combined"]), _stream()),
            "hrs_classes_finish",
        )
        return out

    def merge_classes(self, gathered):
        """gathered int64 [G, B, levels, n]: every shard's nearest n of every class -> [B, levels, n] over all shards."""
        G, B, levels, k = gathered.shape
        return self.merge_topk(gathered.reshape(G, B * levels, k).contiguous()).view(B, levels, k)

    def hybrid_full_corpus(self, corpus, inverted, queries, q_terms, q_term_offs, n, boost=0.3, space="l2", max_terms=None, chunk_bytes=4 << 30,
                           exchange=None):
        """Full-corpus hybrid mode (an extension; the reference's behaviour is hybrid_topk): the n best rows of the
        WHOLE corpus under the fused score of scripts/chromadb_helper.py:302-325 (keyword fraction over term ids),
        ties by (distance, row).  Exact: the best n of every match-count class 0..m by distance contain the best n
        overall.  Queries are processed in chunks whose [chunk, rows] count matrix fits `chunk_bytes`.  On a
        row-sharded corpus `exchange` (keys [B, levels, n] of this shard -> the same over all shards, e.g.
        dist.exchange_classes) sits between the halves; every rank then holds the same result.
        -> dict(rows int64 [B,n], dist fp32 [B,n], counts int32 [B,n], combined fp64 [B,n])."""
        B = queries.shape[0]
        if max_terms is None:
            max_terms = int((q_term_offs[1:] - q_term_offs[:-1]).max().item()) if B else 0
        levels = max_terms + 1
        if levels * n > 256:
            raise ValueError(f"(max query terms + 1) * n = {levels * n} candidates per query exceed 256")
        out = {"rows": [], "dist": [], "counts": [], "combined": []}
        step = max(1, min(B, chunk_bytes // max(1, corpus.n)))
        for b0 in range(0, B, step):
            b1 = min(B, b0 + step)
            qo = (q_term_offs[b0 : b1 + 1] - q_term_offs[b0 : b0 + 1]).contiguous()
            qt = q_terms[int(q_term_offs[b0].item()) :]
            if qt.numel() == 0:
                qt = torch.zeros(1, dtype=torch.int32, device=self.device)
            keys = self.full_corpus_local(corpus, inverted, queries[b0:b1].contiguous(), qt, qo, n, levels, space)
            if exchange is not None:
                keys = exchange(keys)
            part = self.full_corpus_finish(keys, qo, n, boost)
            for k in out:
                out[k].append(part[k])
        return {k: torch.cat(v) for k, v in out.items()}

    def merge_topk(self, keys, vals=None):
        """keys int64(uint64 bits) [G,B,k] (+ vals [G,B,k]: int32, or 64-bit values -- int64 / float64 -- that are moved
        as they are) -> best k per query."""
        G, B, k = keys.shape
        out = torch.empty((B, k), dtype=torch.int64, device=self.device)
        if vals is not None and vals.dtype in (torch.int64, torch.float64):
            ov = torch.empty((B, k), dtype=vals.dtype, device=self.device)
            N.check(N.lib().hrs_merge_topk_i64(self._ctx, _ptr(keys), _ptr(vals.contiguous()), G, B, k, _ptr(out), _ptr(ov), _stream()), "hrs_merge_topk_i64")
            return out, ov
        ov = torch.empty((B, k), dtype=torch.int32, device=self.device) if vals is not None else None
        N.check(N.lib().hrs_merge_topk(self._ctx, _ptr(keys), _ptr(vals), G, B, k, _ptr(out), _ptr(ov), _stream()), "hrs_merge_topk")
        return (out, ov) if vals is not None else out

    def unpack_keys(self, keys):
        dist = torch.empty(keys.shape, dtype=torch.float32, device=self.device)
        rows = torch.empty(keys.shape, dtype=torch.int64, device=self.device)
        N.check(N.lib().hrs_unpack_keys(self._ctx, _ptr(keys), keys.numel(), _ptr(dist), _ptr(rows), _stream()), "hrs_unpack_keys")
        return dist, rows

    def hybrid_topk(self, corpus, postings, queries, q_terms, q_term_offs, n, boost=0.3, space="l2", mode="auto", bm25=None):
        """One pass of the hybrid path for a query batch (scripts/chromadb_helper.py:275-325 batched):
        top-2n by distance -> keyword presence counts on those 2n rows -> fused score, stable order,
        first n.  Returns dict of GPU tensors (rows/dist of the n kept, combined, counts).
        bm25 = dict(idf=fp64 [V] CUDA tensor, avg_len=float[, k1, b]) switches the keyword score from the reference's
        fraction of keywords present to the normalised BM25 weight (an extension): "cand_kw" then holds it."""
        c = 2 * n
        dist, rows = self.score_topk(corpus, queries, c, space=space, mode=mode)
        if bm25 is not None:
            kw = self.bm25_scores(postings, bm25["idf"], bm25["avg_len"], q_terms, q_term_offs, rows, bm25.get("k1", 1.2), bm25.get("b", 0.75))
            combined, perm = self.fuse_sort_scores(dist, kw, n, boost)
            return {"cand_dist": dist, "cand_rows": rows, "cand_kw": kw, "combined": combined, "perm": perm}
        counts = self.keyword_counts(postings, q_terms, q_term_offs, rows)
        combined, perm = self.fuse_sort(dist, counts, q_term_offs, n, boost)
        return {"cand_dist": dist, "cand_rows": rows, "cand_counts": counts, "combined": combined, "perm": perm}

    # ---- synthetic inputs ----------------------------------------------------------------------
    def synth_rows(self, seed, row0, nrows, dim, normalize=True, out=None):
        if out is None:
            out = torch.empty((nrows, dim), dtype=torch.float32, device=self.device)
        N.check(N.lib().hrs_synth_rows(self._ctx, seed, row0, nrows, dim, int(normalize), _ptr(out), _stream()), "hrs_synth_rows")
        return out

    def synth_postings(self, seed, row0, nrows, cdf, L):
        out = torch.empty((nrows, L), dtype=torch.int32, device=self.device)
        N.check(N.lib().hrs_synth_postings(self._ctx, seed, row0, nrows, cdf.numel(), L, _ptr(cdf), _ptr(out), _stream()), "hrs_synth_postings")
        return out
