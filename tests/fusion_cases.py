"""Loads tests/golden/fusion_goldens.json (outputs of the reference's own hybrid_query, written by
tests/golden/make_fusion_goldens.py) as the integer / fp32 arrays the oracle and the HIP kernels take."""
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fusion_goldens.json")


def load():
    """-> list of dicts: doc_offs int64, term_ids int32 (forward CSR of the candidate documents), q_terms int32,
    q_offs int32 [2], rows int64 [1, 2n] (-1 = missing), dist fp32 [1, 2n] (+inf = missing), nvalid, n, boost and
    the reference's order / keyword_score / combined_score."""
    with open(_PATH) as f:
        cases = json.load(f)["cases"]
    out = []
    for c in cases:
        n, c2 = c["n"], 2 * c["n"]
        docs = c["doc_terms"]
        avail = min(c2, len(docs))  # what collection.query(n_results=2n) hands back
        offs = np.zeros(len(docs) + 1, np.int64)
        offs[1:] = np.cumsum([len(d) for d in docs])
        terms = np.array([t for d in docs for t in d], np.int32)
        rows = np.full((1, c2), -1, np.int64)
        rows[0, :avail] = np.arange(avail)
        dist = np.full((1, c2), np.inf, np.float32)
        dist[0, :avail] = np.array(c["distances"][:avail], np.float64).astype(np.float32)
        assert [float(x) for x in dist[0, :avail]] == c["distances"][:avail]  # fp32-representable by construction
        q = np.array(c["q_terms"], np.int32)
        out.append(dict(doc_offs=offs, term_ids=terms, q_terms=q, q_offs=np.array([0, q.size], np.int32), rows=rows, dist=dist,
                        nvalid=np.array([avail], np.int32), n=n, boost=c["boost"], order=c["order"],
                        keyword_score=c["keyword_score"], combined_score=c["combined_score"], semantic_is_int=c["semantic_is_int"]))
    return out


def check(case, counts, combined, perm):
    """counts int32 [1, 2n], combined fp64 [1, n], perm int32 [1, n] against the reference's results, bit for bit."""
    got_n = len(case["order"])
    m = int(case["q_terms"].size)
    assert list(perm[0, :got_n]) == case["order"], (list(perm[0]), case["order"])
    assert (perm[0, got_n:] == -1).all()
    want = np.array(case["combined_score"], np.float64)
    assert np.array_equal(combined[0, :got_n].view(np.uint64), want.view(np.uint64)), (combined[0, :got_n], want)
    kw = [(int(counts[0, j]) / m if m else 0.0) for j in case["order"]]
    assert kw == case["keyword_score"], (kw, case["keyword_score"])
