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
