"""ctypes front end of oracle/search_oracle.c (TEST INFRASTRUCTURE -- see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
SPACES = {"l2": 0, "cosine": 1, "ip": 2}
PAD_KEY = np.uint64(0xFFFFFFFFFFFFFFFF)
_lib = None


def _cpu_has(*flags):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    return all(x in have for x in flags)
    except OSError:
        pass
    return False


def build(force=False):
    """liboracle.so is built with -mavx2 -mfma; a host without them gets a portable build (same results)."""
    cmd = ["make", "-s", "-C", _HERE, "liboracle.so"]
    if not _cpu_has("avx2", "fma"):
        cmd = ["make", "-s", "-B", "-C", _HERE, "liboracle.so", "ARCH="]
    elif force:
        cmd.insert(2, "-B")
    subprocess.check_call(cmd)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or not _cpu_has("avx2", "fma"):
            build()
        L = C.CDLL(_SO)
        L.hrs_oracle_dist64.restype = C.c_double
        L.hrs_oracle_dist64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def dist64(q, c, space="l2"):
    q = np.ascontiguousarray(q, np.float32)
    c = np.ascontiguousarray(c, np.float32)
    return lib().hrs_oracle_dist64(_p(q), _p(c), C.c_int(q.shape[0]), C.c_int(SPACES[space]))


def topk_keys(Q, Cm, k, space="l2", row_offset=0, mask=None, impl=None):
    """impl: None picks by batch size; "plain" (one query per thread), "rows" (threads share the rows of one
    query) and "blocked" (eight queries per pass) give bit-identical keys."""
    Q = np.ascontiguousarray(Q, np.float32)
    Cm = np.ascontiguousarray(Cm, np.float32)
    B, D = Q.shape
    N = Cm.shape[0]
    out = np.empty((B, k), np.uint64)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
    if impl is None:
        impl = "blocked" if B >= 128 else ("plain" if B >= 4 else "rows")
    fn = {"plain": lib().hrs_oracle_topk, "rows": lib().hrs_oracle_topk_rows, "blocked": lib().hrs_oracle_topk_blocked}[impl]
    fn(_p(Q), _p(Cm), C.c_int64(N), C.c_int(D), C.c_int(B), C.c_int(k), C.c_int(SPACES[space]), C.c_int64(row_offset), _p(mask), _p(out))
    return out


def unpack(keys):
    keys = np.ascontiguousarray(keys, np.uint64)
    dist = np.empty(keys.shape, np.float32)
    rows = np.empty(keys.shape, np.int64)
    lib().hrs_oracle_unpack(_p(keys), C.c_int64(keys.size), _p(dist), _p(rows))
    return dist, rows


def topk(Q, Cm, k, space="l2", row_offset=0, mask=None):
    """-> (distances fp32 [B,k], rows int64 [B,k]); missing entries are (+inf, -1)."""
    return unpack(topk_keys(Q, Cm, k, space, row_offset, mask))


def merge(keys):
    keys = np.ascontiguousarray(keys, np.uint64)
    G, B, k = keys.shape
    out = np.empty((B, k), np.uint64)
    lib().hrs_oracle_merge(_p(keys), C.c_int(G), C.c_int(B), C.c_int(k), _p(out))
    return out


def keyword_counts(doc_offs, term_ids, q_term_offs, q_terms, rows, row_offset=0):
    doc_offs = np.ascontiguousarray(doc_offs, np.int64)
    term_ids = np.ascontiguousarray(term_ids, np.int32)
    q_term_offs = np.ascontiguousarray(q_term_offs, np.int32)
    q_terms = np.ascontiguousarray(q_terms, np.int32)
    rows = np.ascontiguousarray(rows, np.int64)
    B, c = rows.shape
    counts = np.empty((B, c), np.int32)
    lib().hrs_oracle_keyword_counts(_p(doc_offs), _p(term_ids), _p(q_term_offs), _p(q_terms), _p(rows), C.c_int(B), C.c_int(c), C.c_int64(row_offset), _p(counts))
    return counts


def fuse_sort(dist, counts, n_terms, n, boost=0.3, nvalid=None):
    """-> combined fp64 [B,n], perm int32 [B,n], sem fp64 [B,c], kw fp64 [B,c]."""
    dist = np.ascontiguousarray(dist, np.float32)
    counts = np.ascontiguousarray(counts, np.int32)
    n_terms = np.ascontiguousarray(n_terms, np.int32)
    B, c = dist.shape
    if nvalid is not None:
        nvalid = np.ascontiguousarray(nvalid, np.int32)
    comb = np.empty((B, n), np.float64)
    perm = np.empty((B, n), np.int32)
    sem = np.zeros((B, c), np.float64)
    kw = np.zeros((B, c), np.float64)
    lib().hrs_oracle_fuse_sort(_p(dist), _p(counts), _p(n_terms), _p(nvalid), C.c_int(B), C.c_int(c), C.c_double(boost), C.c_int(n), _p(comb), _p(perm), _p(sem), _p(kw))
    return comb, perm, sem, kw


def hybrid_full(Q, Cm, doc_offs, term_ids, q_term_offs, q_terms, n, boost=0.3, space="l2"):
    """Full-corpus fusion by brute force (an extension of the reference's re-rank; parity unpinned)
    -> rows int64 [B,n], combined fp64 [B,n], counts int32 [B,n], dist fp32 [B,n]."""
    Q = np.ascontiguousarray(Q, np.float32)
    Cm = np.ascontiguousarray(Cm, np.float32)
    doc_offs = np.ascontiguousarray(doc_offs, np.int64)
    term_ids = np.ascontiguousarray(term_ids, np.int32)
    q_term_offs = np.ascontiguousarray(q_term_offs, np.int32)
    q_terms = np.ascontiguousarray(q_terms, np.int32)
    B, D = Q.shape
    rows = np.empty((B, n), np.int64)
    comb = np.empty((B, n), np.float64)
    counts = np.empty((B, n), np.int32)
    dist = np.empty((B, n), np.float32)
    lib().hrs_oracle_hybrid_full(_p(Q), _p(Cm), C.c_int64(Cm.shape[0]), C.c_int(D), C.c_int(B), C.c_int(SPACES[space]), _p(doc_offs), _p(term_ids),
                                 _p(q_term_offs), _p(q_terms), C.c_int(n), C.c_double(boost), _p(rows), _p(comb), _p(counts), _p(dist))
    return rows, comb, counts, dist


def bm25_idf(doc_freq, n_docs):
    """idf of the BM25 extension: ln(1 + (N - df + 0.5) / (df + 0.5)), numpy float64 (its own definition; no reference)."""
    df = np.asarray(doc_freq, np.float64)
    return np.log(1.0 + (float(n_docs) - df + 0.5) / (df + 0.5))


def bm25_scores(doc_offs, term_ids, idf, avg_len, q_term_offs, q_terms, rows, k1=1.2, b=0.75, row_offset=0):
    """BM25 keyword score of every (query, candidate) pair, normalised to [0, 1) (an extension of
    scripts/chromadb_helper.py:352-359, which counts presence; own semantics, parity unpinned).  Python floats: every
    operation is one rounded fp64 operation, summed in query-term order.  rows int64 [B, c] -> fp64 [B, c]."""
    doc_offs = np.asarray(doc_offs, np.int64)
    term_ids = np.asarray(term_ids, np.int32)
    idf = np.asarray(idf, np.float64)
    rows = np.asarray(rows, np.int64)
    B, c = rows.shape
    out = np.zeros((B, c), np.float64)
    n_rows = len(doc_offs) - 1
    k1, b, avg_len = float(k1), float(b), float(avg_len)
    for q in range(B):
        terms = [int(t) for t in q_terms[q_term_offs[q] : q_term_offs[q + 1]]]
        for j in range(c):
            r = int(rows[q, j]) - row_offset
            if rows[q, j] < 0 or r < 0 or r >= n_rows:
                continue
            lst = term_ids[doc_offs[r] : doc_offs[r + 1]].tolist()
            norm = k1 * ((1.0 - b) + b * (float(len(lst)) / avg_len))
            score = best = 0.0
            for t in terms:
                if t < 0 or t >= len(idf):
                    continue
                w = float(idf[t])
                best = best + w * (k1 + 1.0)
                f = lst.count(t)
                if f > 0:
                    score = score + ((w * float(f)) * (k1 + 1.0)) / (float(f) + norm)
            out[q, j] = score / best if best > 0.0 else 0.0
    return out


def fuse_sort_scores(dist, kw, n, boost=0.3):
    """The fusion and stable order of scripts/chromadb_helper.py:302-325 with the keyword score given per candidate
    (Python floats) -> combined fp64 [B, n] (-1 where missing), perm int32 [B, n] (-1)."""
    dist = np.asarray(dist, np.float32)
    B, c = dist.shape
    comb = np.full((B, n), -1.0, np.float64)
    perm = np.full((B, n), -1, np.int32)
    for q in range(B):
        scored = []
        for j in range(c):
            if np.isinf(dist[q, j]):
                continue
            sem = max(0.0, 1.0 - float(dist[q, j]))
            scored.append((sem * (1.0 - boost) + float(kw[q, j]) * boost, j))
        scored.sort(key=lambda x: x[0], reverse=True)  # stable
        for i, (v, j) in enumerate(scored[:n]):
            comb[q, i], perm[q, i] = v, j
    return comb, perm
