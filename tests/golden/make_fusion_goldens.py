#!/usr/bin/env python3
"""Generate tests/golden/fusion_goldens.json by running the REFERENCE's own hybrid_query.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fusion_goldens.py

Purpose: pin the integer keyword-count + fp64 fusion + stable-order arithmetic (the C oracle's
hrs_oracle_keyword_counts / hrs_oracle_fuse_sort and the HIP kernels behind hrs_keyword_counts /
hrs_fuse_sort) to the reference's scripts/chromadb_helper.py:296-325 and 352-359 themselves.

The reference tests "keyword in document" on strings.  Here every integer term id t is written as a
fixed-width lower-case code (`code(t)`: the letter q followed by five base-26 letters -- never a stop
word, longer than two letters, matched whole by the reference's r'\\b[a-z]+\\b'), and a document is the
codes of its terms joined by single spaces.  All codes have the same width and are space-separated, so a
code occurs as a substring of a document exactly when the term is in the document's list: the
reference's substring test IS term membership on these inputs, and extract_keywords(query) returns
exactly the query's distinct codes.  Distances are fp32-representable (what ChromaDB returns).

The output holds inputs (term lists, distances) and the reference's outputs only.
"""
import json
import os
import sys

import numpy as np

REF = os.environ.get("HRS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "scripts"))

import chromadb_helper as ref  # noqa: E402


def code(t):
    s = ""
    for _ in range(5):
        s = chr(ord("a") + t % 26) + s
        t //= 26
    return "q" + s


class Stub:
    name = "stub"
    metadata = {}

    def __init__(self, docs, dists):
        self.docs, self.dists = docs, dists
        self.n_results = None

    def query(self, query_texts, n_results, include, where=None):
        self.n_results = n_results
        k = min(n_results, len(self.docs))
        return {
            "ids": [[str(i) for i in range(k)]],
            "documents": [self.docs[:k]],
            "metadatas": [[{} for _ in range(k)]],
            "distances": [self.dists[:k]],
        }


def f32(x):
    return float(np.float32(x))


def run_case(q_terms, extra_words, doc_terms, dists, n, boost):
    """-> the case record: inputs + what the reference returned."""
    docs = [" ".join(code(t) for t in terms) for terms in doc_terms]
    qtext = " ".join([code(t) for t in q_terms] + list(extra_words))
    stub = Stub(docs, dists)
    ref.get_collection = lambda name=None, _s=stub: _s
    res = ref.hybrid_query(qtext, n, "c", boost)
    kws = ref.extract_keywords(qtext)
    assert sorted(kws) == sorted({code(t) for t in q_terms}), (kws, q_terms)
    assert stub.n_results == 2 * n
    return dict(
        q_terms=sorted(set(int(t) for t in q_terms)),
        doc_terms=[[int(t) for t in terms] for terms in doc_terms],
        distances=[float(d) for d in dists],
        n=n,
        boost=boost,
        order=[int(r["id"]) for r in res],
        keyword_score=[r["keyword_score"] for r in res],
        semantic_score=[float(r["semantic_score"]) for r in res],
        semantic_is_int=[isinstance(r["semantic_score"], int) for r in res],
        combined_score=[r["combined_score"] for r in res],
    )


def main():
    rng = np.random.default_rng(20260)
    cases = []
    # hand-made corners: clamp (d > 1 -> int 0), d == 1, exact ties, fewer rows than 2n, no rows, no keywords,
    # duplicate query terms, an empty document, a term beyond every list
    t3 = [5, 17, 123456]
    cases.append(run_case(t3, [], [[5, 17, 99], [5], [], [17, 123456, 5], [1, 2, 3]], [f32(0.25), f32(0.25), f32(0.5), f32(1.0), f32(1.75)], 5, 0.3))
    cases.append(run_case(t3, ["the", "and", "of"], [[5, 17, 99], [5], [], [17, 123456, 5]], [f32(0.1)] * 4, 2, 0.3))
    cases.append(run_case([], ["the", "and", "of", "to"], [[5], [6], [7]], [f32(0.2), f32(0.4), f32(0.6)], 3, 0.3))
    cases.append(run_case([9, 9, 9, 4], [], [[9], [4], [4, 9], [8]], [f32(0.9), f32(0.9), f32(1.1), f32(1.1)], 1, 0.3))
    cases.append(run_case([7], [], [], [], 4, 0.3))
    cases.append(run_case([11881375], [], [[0], [11881375]], [f32(1e-7), f32(2.0)], 10, 0.3))  # the largest 5-letter code
    cases.append(run_case([1, 2, 3, 4, 5, 6, 7], [], [[1, 2, 3, 4, 5, 6, 7], [1, 3, 5, 7], [2, 4, 6], [1], []], sorted([f32(0.999999), f32(1.0), f32(1.0000001), f32(0.3), f32(0.3)]), 3, 0.3))
    # seeded random cases: ragged term lists, 0..8 query terms, every kind of distance, several boosts
    special = [0.0, 1.0, 0.5, 0.75, 1.0000001192092896, 0.9999999403953552, 2.0, 1e-7, 3.5]
    for ci in range(60):
        V = int(rng.choice([12, 40, 400]))
        n = int(rng.choice([1, 2, 3, 5, 10, 20]))
        avail = int(rng.choice([2 * n, 2 * n, 2 * n + 3, max(0, 2 * n - 3), n]))
        m = int(rng.integers(0, 9))
        q_terms = [int(t) for t in rng.integers(0, V, m)]  # duplicates allowed: the reference de-duplicates
        doc_terms = []
        for _ in range(avail):
            L = int(rng.integers(0, min(V, 30) + 1))
            doc_terms.append(sorted(int(t) for t in rng.choice(V, L, replace=False)))
        d = rng.uniform(0.0, 2.2, avail)
        pick = rng.random(avail) < 0.3
        d[pick] = rng.choice(special, int(pick.sum()))
        if avail > 3 and ci % 3 == 0:
            d[1:4] = d[1]  # a run of exact ties
        dists = sorted(f32(x) for x in d)
        boost = [0.3, 0.3, 0.3, 0.5, 0.0, 1.0, 0.25][ci % 7]
        extra = ["the", "with"] if ci % 5 == 0 else []
        cases.append(run_case(q_terms, extra, doc_terms, dists, n, boost))
    path = os.path.join(HERE, "fusion_goldens.json")
    with open(path, "w") as f:
        json.dump(dict(cases=cases), f, separators=(",", ":"))
    print("wrote", path, len(cases), "cases;", sum(len(c["order"]) for c in cases), "result rows;",
          sum(any(c["semantic_is_int"]) for c in cases), "cases with a clamped score")


if __name__ == "__main__":
    main()
