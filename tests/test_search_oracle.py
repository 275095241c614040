"""The C search oracle: against straightforward numpy fp64, its stored vectors, and its own edge cases."""
import importlib.util
import os

import numpy as np

from oracle import search_oracle as so

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load_inputs():
    spec = importlib.util.spec_from_file_location("msg", os.path.join(GOLD, "make_search_goldens.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.inputs()


def test_golden_vectors():
    Q, C = _load_inputs()
    gold = np.load(os.path.join(GOLD, "search_goldens.npz"))
    for name in gold.files:
        space, k = name.split("_")
        assert np.array_equal(so.topk_keys(Q, C, int(k), space), gold[name]), name
    # the duplicate rows 7, 100, 4000 tie exactly and come back in row order
    d, r = so.unpack(gold["l2_10"])
    assert list(r[0, :3]) == [7, 100, 4000] and (d[0, :3] == 0).all()


def test_against_numpy_fp64():
    Q, C = _load_inputs()
    Qd, Cd = Q.astype(np.float64), C.astype(np.float64)
    dot = Qd @ Cd.T
    ref = {
        "l2": ((Qd[:, None, :] - Cd[None, :, :]) ** 2).sum(-1),
        "ip": 1 - dot,
        "cosine": 1 - dot / (np.linalg.norm(Qd, axis=1)[:, None] * np.linalg.norm(Cd, axis=1)[None, :]),
    }
    for space, full in ref.items():
        d, r = so.topk(Q, C, 10, space)
        # same set and distances within fp32 rounding (summation order differs from numpy's)
        want = np.argsort(full, axis=1, kind="stable")[:, :10]
        assert (np.sort(want, 1) == np.sort(r, 1)).mean() > 0.99
        assert np.abs(np.take_along_axis(full, r, 1) - d).max() < 5e-7
        assert (np.diff(d, axis=1) >= 0).all()


def test_row_parallel_variant_and_mask_and_offset():
    Q, C = _load_inputs()
    a = so.topk_keys(Q[:2], C, 10, "l2")  # B < 4 -> row-parallel code path
    b = so.topk_keys(Q, C, 10, "l2")[:2]
    assert np.array_equal(a, b)
    mask = np.zeros(C.shape[0], np.uint8)
    mask[::3] = 1
    d, r = so.topk(Q, C, 5, "l2", mask=mask)
    assert (r % 3 == 0).all()
    d2, r2 = so.topk(Q, C, 5, "l2", row_offset=10_000)
    d1, r1 = so.topk(Q, C, 5, "l2")
    assert np.array_equal(r2, r1 + 10_000) and np.array_equal(d1, d2)
    # fewer rows than k
    d3, r3 = so.topk(Q[:1], C[:3], 5, "l2")
    assert list(r3[0, 3:]) == [-1, -1] and np.isinf(d3[0, 3:]).all()


def test_merge_equals_unsharded():
    Q, C = _load_inputs()
    want = so.topk_keys(Q, C, 20, "l2")
    for G in (2, 4, 8):
        b = [C.shape[0] * g // G for g in range(G + 1)]
        parts = np.stack([so.topk_keys(Q, C[b[g] : b[g + 1]], 20, "l2", row_offset=b[g]) for g in range(G)])
        assert np.array_equal(so.merge(parts), want)


def test_fuse_sort_matches_python_floats():
    from oracle import text_oracle as T

    rng = np.random.default_rng(3)
    B, c, n = 9, 10, 5
    dist = np.sort(rng.uniform(0, 2.5, (B, c)).astype(np.float32), axis=1)
    dist[0] = 0.5
    counts = rng.integers(0, 5, (B, c)).astype(np.int32)
    nt = rng.integers(0, 5, B).astype(np.int32)
    counts = np.minimum(counts, nt[:, None])
    comb, perm, sem, kw = so.fuse_sort(dist, counts, nt, n, 0.3)
    for b in range(B):
        rows = []
        for j in range(c):
            d = float(dist[b, j])
            s = max(0, 1 - d)
            k = counts[b, j] / nt[b] if nt[b] else 0.0
            rows.append((s * (1 - 0.3) + k * 0.3, j))
        rows.sort(key=lambda t: t[0], reverse=True)
        assert [j for _, j in rows[:n]] == list(perm[b])
        assert [v for v, _ in rows[:n]] == list(comb[b])
    assert T.keyword_score("abc docker", ["docker", "zzz"]) == 0.5


def test_fusion_and_keyword_counts_reproduce_the_reference():
    """hrs_oracle_keyword_counts + hrs_oracle_fuse_sort against what the reference's own hybrid_query returned
    (scripts/chromadb_helper.py:296-325, 352-359) on documents whose text spells integer term lists."""
    import fusion_cases

    cases = fusion_cases.load()
    assert len(cases) >= 60 and any(any(c["semantic_is_int"]) for c in cases)
    for c in cases:
        counts = so.keyword_counts(c["doc_offs"], c["term_ids"], c["q_offs"], c["q_terms"], c["rows"])
        comb, perm, _, _ = so.fuse_sort(c["dist"], counts, np.diff(c["q_offs"]), c["n"], c["boost"], c["nvalid"])
        fusion_cases.check(c, counts, comb, perm)


def test_blocked_and_plain_scans_give_identical_keys():
    rng = np.random.default_rng(8)
    n, d, b, k = 3001, 100, 70, 20
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((b, d)).astype(np.float32)
    C[5] = 0
    Q[3] = 0
    C[100] = C[7]
    Q[9] = C[7]
    mask = (rng.random(n) < 0.7).astype(np.uint8)
    for space in ("l2", "cosine", "ip"):
        for m in (None, mask):
            a = so.topk_keys(Q, C, k, space, row_offset=17, mask=m, impl="plain")
            assert np.array_equal(a, so.topk_keys(Q, C, k, space, row_offset=17, mask=m, impl="blocked")), space
            assert np.array_equal(a[:3], so.topk_keys(Q[:3], C, k, space, row_offset=17, mask=m, impl="rows")), space
