"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest
import torch

from oracle import search_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from hybrid_rag_system_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def _keys(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
@pytest.mark.parametrize("n,d,b,k", [(20037, 384, 5, 10), (1000, 100, 3, 7), (4099, 1024, 2, 100), (63, 384, 1, 10)])
def test_exact_scan_bit_exact(eng, space, n, d, b, k):
    rng = np.random.default_rng(n + d)
    C = _unit(rng, n, d)
    if space != "l2":
        C *= rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    C[n // 2] = C[3]  # exact duplicates: tie broken by row
    C[n - 1] = C[3]
    Q = _unit(rng, b, d)
    Q[0] = C[3]
    want = so.topk_keys(Q, C, k, space)
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    dist, rows, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, space=space, mode="exact", want_keys=True)
    torch.cuda.synchronize()
    assert np.array_equal(_keys(keys), want)
    wd, wr = so.unpack(want)
    assert np.array_equal(rows.cpu().numpy(), wr)
    assert np.array_equal(dist.cpu().numpy(), wd)
    corpus.close()


def test_exact_fewer_rows_than_k_and_offset(eng):
    rng = np.random.default_rng(5)
    C = _unit(rng, 6, 384)
    Q = _unit(rng, 2, 384)
    want = so.topk_keys(Q, C, 10, "l2", row_offset=1000)
    corpus = eng.corpus(torch.from_numpy(C).cuda(), row_offset=1000)
    dist, rows, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), 10, want_keys=True)
    torch.cuda.synchronize()
    assert np.array_equal(_keys(keys), want)
    assert (rows.cpu().numpy()[:, 6:] == -1).all() and np.isinf(dist.cpu().numpy()[:, 6:]).all()
    assert rows.cpu().numpy()[:, :6].min() >= 1000
    corpus.close()


def test_exact_mask(eng):
    rng = np.random.default_rng(6)
    C = _unit(rng, 5000, 384)
    Q = _unit(rng, 4, 384)
    mask = (rng.random(5000) < 0.3).astype(np.uint8)
    want = so.topk_keys(Q, C, 10, "l2", mask=mask)
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    _, _, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), 10, mask=torch.from_numpy(mask).cuda(), want_keys=True)
    torch.cuda.synchronize()
    assert np.array_equal(_keys(keys), want)
    corpus.close()


@pytest.mark.parametrize("keep", [0.5, 0.02])
def test_mfma_with_row_mask(eng, keep):
    """The `where` filter on the matrix-core path: masked rows are never recorded; a mask that keeps few
    rows sends queries to the (masked) exact scan.  Same keys as the oracle either way."""
    rng = np.random.default_rng(8)
    n, d, b, k = 100_003, 384, 40, 10
    C = _unit(rng, n, d)
    Q = _unit(rng, b, d)
    for j in range(10):
        Q[j] = C[j * 4999]  # some of these rows are masked out: their duplicates must not come back
    mask = (rng.random(n) < keep).astype(np.uint8)
    want = so.topk_keys(Q, C, k, "l2", mask=mask)
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    _, rows, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, mask=torch.from_numpy(mask).cuda(), mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    assert np.array_equal(_keys(keys), want), st
    r = rows.cpu().numpy()
    assert (mask[r[r >= 0]] == 1).all()
    if keep == 0.5:
        assert st["fallback_queries"] <= 3, st
    corpus.close()


def test_empty_corpus(eng):
    corpus = eng.corpus(torch.empty((0, 384), dtype=torch.float32, device="cuda"))
    dist, rows = eng.score_topk(corpus, torch.zeros((2, 384), device="cuda"), 5)
    torch.cuda.synchronize()
    assert (rows.cpu().numpy() == -1).all()
    corpus.close()


@pytest.mark.parametrize("shadow", ["i8", "bf16"])
@pytest.mark.parametrize("space", ["l2", "cosine", "ip"])
def test_mfma_scan_bit_exact(eng, space, shadow):
    """Matrix-core path (int8 and bf16 shadows): same keys as the exact oracle (ids, order and fp32 distances)."""
    rng = np.random.default_rng(11)
    n, d, b, k = 150_001, 384, 300, 20
    C = _unit(rng, n, d)
    Q = _unit(rng, b, d)
    # planted near-duplicates (semantic score > 0 so the clamp is exercised later) and exact duplicates
    for j in range(0, 40):
        Q[j] = C[j * 997] + 0.05 * rng.standard_normal(d).astype(np.float32) / np.sqrt(d)
        Q[j] /= np.linalg.norm(Q[j])
    C[77777] = C[0]
    C[150_000] = C[0]
    want = so.topk_keys(Q, C, k, space)
    corpus = eng.corpus(torch.from_numpy(C).cuda(), shadow=shadow)
    dist, rows, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, space=space, mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    got = _keys(keys)
    assert np.array_equal(got, want), f"mismatching queries: {np.nonzero((got != want).any(1))[0][:10]} stats={st}"
    assert st["fallback_queries"] <= 3, st
    assert st["overflow_queries"] == 0 and st["max_candidates"] < 4096, st
    corpus.close()


@pytest.mark.parametrize("space", ["ip", "l2"])
@pytest.mark.parametrize("d", [384, 1024])
def test_int8_bound_holds_when_the_quantisation_residuals_align(eng, d, space):
    """The int8 scan drops a row only if an upper bound of its score is below the threshold; the bound takes the
    three cross terms of the quantisation residuals by Cauchy-Schwarz, which random residuals never come near.
    Here they are made to: every row is (integer vector + 7/16 sign(q)) x 2^-7, so its residual is parallel to the
    integer vector of the first queries (all +-127) and the bound is met with equality up to its rounding margin;
    the other queries carry a residual parallel to one row's integer vector.  Were any norm in the bound a little
    too small (a rounding the wrong way, a lane left out of a reduction), the rows that ought to win would score
    above their bound and could be dropped; the keys must equal the oracle's."""
    rng = np.random.default_rng(5 + d)
    n, b, k = 20_000, 8, 10
    sign = rng.choice([-1.0, 1.0], d).astype(np.float32)
    ci = rng.integers(-100, 101, (n, d)).astype(np.float32)
    ci[::32, 0] = 127.0  # every 32-row group's scale is exactly 2^-7
    C = ((ci + np.float32(0.4375) * sign) * np.float32(2.0**-7)).astype(np.float32)
    C[::32, 0] = 127.0 * 2.0**-7
    Q = np.empty((b, d), np.float32)
    Q[:4] = sign * np.float32(2.0**-3)                      # integer vector +-127, no residual
    Q[1] *= 3.0
    Q[2, : d // 2] *= 0.5                                     # (half the components at +-64 after quantisation)
    Q[3] = -Q[3]
    for j in range(4, b):                                     # integer part random, residual parallel to a row's integers
        qi = rng.integers(-126, 127, d).astype(np.float32)
        qi[0] = 127.0
        row = ci[1 + 37 * j]
        Q[j] = (qi + np.float32(0.4375) * np.sign(row)) * np.float32(2.0**-5)
        Q[j, 0] = 127.0 * 2.0**-5
    want = so.topk_keys(Q, C, k, space)
    corpus = eng.corpus(torch.from_numpy(C).cuda(), shadow="i8")
    _, _, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, space=space, mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    assert np.array_equal(_keys(keys), want), st
    if space == "ip":
        assert st["fallback_queries"] == 0, st  # answered by the int8 scan and its proof, not by the exact fallback
    corpus.close()


@pytest.mark.parametrize("shadow", ["i8", "bf16"])
def test_mfma_padding_and_zero_queries_record_nothing(eng, shadow):
    """A partial batch is padded with queries of zeros, and a caller may send one: their integer scores are all 0,
    nearer to any threshold than the int8 pass's integer compare resolves, so without a threshold out of reach every
    row of the corpus is recorded for each of them (a batch of 4 over 10M rows took 115 ms instead of 0.8).  Checked
    by results (equal to the exact scan; the query of zeros is answered by the fallback) and by the clock, with a
    margin of more than ten times either way."""
    n, d, k = 2_000_000, 384, 10
    m = torch.empty((n, d), dtype=torch.float32, device="cuda")
    for r0 in range(0, n, 1 << 19):
        nr = min(1 << 19, n - r0)
        eng.synth_rows(99, r0, nr, d, out=m[r0 : r0 + nr])
    corpus = eng.corpus(m, shadow=shadow)
    q = eng.synth_rows(7, 0, 6, d)
    q[3] = 0
    for space in ("cosine", "l2"):
        _, _, want = eng.score_topk(corpus, q, k, space=space, mode="exact", want_keys=True)
        _, _, keys = eng.score_topk(corpus, q, k, space=space, mode="mfma", want_keys=True)
        torch.cuda.synchronize()
        st = eng.score_stats()
        assert torch.equal(keys, want), (space, st)
        assert st["fallback_queries"] <= 1 and st["overflow_queries"] == 0, (space, st)
    q4 = q[[0, 1, 2, 4]].contiguous()
    eng.score_topk(corpus, q4, k, space="cosine", mode="mfma")
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        eng.score_topk(corpus, q4, k, space="cosine", mode="mfma")
    t1.record()
    torch.cuda.synchronize()
    assert eng.score_stats()["fallback_queries"] == 0
    assert t0.elapsed_time(t1) / 5 < 4.0, "a batch of 4 over 2M rows takes ~0.3 ms; 20+ ms when the padding records rows"
    corpus.close()


@pytest.mark.parametrize("shadow", ["i8", "bf16"])
@pytest.mark.parametrize("d,b,space", [(1024, 200, "cosine"), (1024, 40, "l2"), (768, 130, "ip"), (512, 70, "l2"), (100, 300, "cosine")])
def test_mfma_scan_wide_bit_exact(eng, d, b, space, shadow):
    """Other depths: every k-step count of both shadows (bf16 above 512: the four-wave kernel with
    AGPR-resident query fragments), all query-group counts."""
    rng = np.random.default_rng(d + b)
    n, k = 90_003, 10
    C = _unit(rng, n, d)
    if space == "ip":
        C *= rng.uniform(0.7, 1.3, (n, 1)).astype(np.float32)
    Q = _unit(rng, b, d)
    for j in range(0, 20):
        Q[j] = C[j * 1999] + 0.05 * rng.standard_normal(d).astype(np.float32) / np.sqrt(d)
        Q[j] /= np.linalg.norm(Q[j])
    C[n - 1] = C[0]
    want = so.topk_keys(Q, C, k, space)
    corpus = eng.corpus(torch.from_numpy(C).cuda(), shadow=shadow)
    _, _, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, space=space, mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    got = _keys(keys)
    assert np.array_equal(got, want), f"mismatching queries: {np.nonzero((got != want).any(1))[0][:10]} stats={st}"
    # the int8 error bound is wide next to the score spread of deep random rows: those queries take the
    # exact fallback (same results); AUTO therefore builds the bf16 shadow above depth 512
    if not (shadow == "i8" and d > 512):
        assert st["fallback_queries"] <= 3, st
    assert st["overflow_queries"] == 0, st
    corpus.close()


@pytest.mark.parametrize("shadow", ["i8", "bf16"])
@pytest.mark.parametrize("b,space", [(1024, "cosine"), (640, "l2")])
def test_mfma_scan_wide_large_batch_top100(eng, b, space, shadow):
    """The shape of a 100M x 1024 corpus' shard query (batch 1024, top-100) at a size the oracle finishes in seconds,
    and every key equals the oracle's.  bf16 shadow: batches of 512 and more take the one-barrier-per-tile schedule of
    the depth-1024 kernel (eight resp. five query groups of 128).  int8 shadow (the default at this depth): the 32-step
    int8 kernel, lists of ~2700 rows, every query re-scored by the large variant of the re-score kernel."""
    rng = np.random.default_rng(1000 + b)
    n, d, k = 120_007, 1024, 100
    C = _unit(rng, n, d)  # unit rows, as embeddings are: the cosine proof bounds a row's norm by the corpus' extremes
    Q = _unit(rng, b, d)
    for j in range(0, 64):  # near neighbours of corpus rows, in every query group
        qi = (j * 16) % b
        Q[qi] = C[j * 1801] + 0.3 * rng.standard_normal(d).astype(np.float32) / np.sqrt(d)
        Q[qi] /= np.linalg.norm(Q[qi])
    C[n - 1] = C[0]
    C[60_000] = C[0]
    Q[b - 1] = C[0] / np.linalg.norm(C[0])
    want = so.topk_keys(Q, C, k, space)
    corpus = eng.corpus(torch.from_numpy(C).cuda(), shadow=shadow)
    _, _, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, space=space, mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    got = _keys(keys)
    assert np.array_equal(got, want), f"mismatching queries: {np.nonzero((got != want).any(1))[0][:10]} stats={st}"
    assert st["overflow_queries"] == 0 and st["fallback_queries"] <= b // 50, st
    # the same call in two halves (what the pipelined multi-GPU step issues)
    eng.score_topk_begin(corpus, torch.from_numpy(Q).cuda(), k, space=space, mode="mfma")
    _, _, keys2 = eng.score_topk_finish(want_keys=True)
    torch.cuda.synchronize()
    assert np.array_equal(_keys(keys2), want)
    corpus.close()


@pytest.mark.parametrize("shadow", ["i8", "bf16"])
@pytest.mark.parametrize("space", ["l2", "ip", "cosine"])
def test_mfma_awkward_rows(eng, space, shadow):
    """Rows the shadows quantise badly or not at all: zero rows and a zero group, a zero query, outliers
    that coarsen their group's int8 scale, all-negative scores, a ragged last tile.  Results stay exact
    (the exact fallback may answer some of these queries)."""
    rng = np.random.default_rng(99)
    n, d, b, k = 20_011, 192, 12, 10
    C = _unit(rng, n, d)
    C[64:96] = 0.0                       # a whole 32-row group of zeros
    C[500] = 0.0                         # a single zero row
    C[1000] *= 300.0                     # outliers: one huge row, one huge element
    C[2000, 7] = 40.0
    C[3000:3200] = np.abs(C[3000:3200])  # rows in the positive orthant
    C[n - 1] = C[5]                      # duplicate in the ragged last tile
    Q = _unit(rng, b, d)
    Q[0] = 0.0                           # zero query
    Q[1] = -np.abs(Q[1])                 # negative scores against the positive-orthant rows
    Q[2] = C[1000] / np.linalg.norm(C[1000])
    Q[3] = C[5]
    Q[4] *= 1e-3                         # tiny and huge query norms
    Q[5] *= 1e3
    want = so.topk_keys(Q, C, k, space)
    corpus = eng.corpus(torch.from_numpy(C).cuda(), shadow=shadow)
    _, _, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, space=space, mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    got = _keys(keys)
    assert np.array_equal(got, want), f"mismatching queries: {np.nonzero((got != want).any(1))[0][:10]} stats={st}"
    corpus.close()


@pytest.mark.parametrize("shadow", ["i8", "bf16"])
def test_mfma_topically_ordered_corpus(eng, shadow):
    """Rows that come in runs of near neighbours (documents indexed file by file): scores are not
    independent, whole runs exceed a threshold together.  The sampled ladder count raises the threshold
    so that the candidate lists do not overflow, and the results stay exact."""
    rng = np.random.default_rng(5)
    d, csize, ncl = 384, 500, 160
    centers = _unit(rng, ncl, d)
    C = np.repeat(centers, csize, axis=0) + (0.3 / np.sqrt(d)) * rng.standard_normal((ncl * csize, d)).astype(np.float32)
    C = (C / np.linalg.norm(C, axis=1, keepdims=True)).astype(np.float32)
    Q = centers[rng.integers(0, ncl, 48)] + (0.1 / np.sqrt(d)) * rng.standard_normal((48, d)).astype(np.float32)
    Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    want = so.topk_keys(Q, C, 10, "l2")
    corpus = eng.corpus(torch.from_numpy(C).cuda(), shadow=shadow)
    _, _, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), 10, mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    assert np.array_equal(_keys(keys), want), st
    assert st["overflow_queries"] == 0 and st["fallback_queries"] <= 2, st
    corpus.close()


@pytest.mark.parametrize("shadow", ["i8", "bf16"])
def test_mfma_fallback_on_heavy_ties(eng, shadow):
    """More exact duplicates of the best row than the largest re-score window (4096): the proof fails and
    the query is answered by the exact scan, still bit-exact.  Two more queries have more duplicates than
    the first window (256) but fewer than the fast kernel's largest (1024), resp. more than that but fewer
    than the deferred large kernel's: they are proven after the window has grown."""
    rng = np.random.default_rng(12)
    n, d, b, k = 70_000, 128, 20, 10
    C = _unit(rng, n, d)
    C[1000:6200] = C[5]
    C[40000:40400] = C[7]
    C[50000:51600] = C[9]
    Q = _unit(rng, b, d)
    Q[3] = C[5]
    Q[4] = C[7]
    Q[6] = C[9]
    want = so.topk_keys(Q, C, k, "l2")
    corpus = eng.corpus(torch.from_numpy(C).cuda(), shadow=shadow)
    _, _, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    assert np.array_equal(_keys(keys), want), st
    assert 1 <= st["fallback_queries"] <= 2, st
    corpus.close()


def _postings(rng, n, V, L):
    terms = np.empty((n, L), np.int32)
    for i in range(n):
        terms[i] = np.sort(rng.choice(V, L, replace=False))
    return np.arange(n + 1, dtype=np.int64) * L, terms.reshape(-1)


def test_keyword_counts_fuse_merge(eng):
    rng = np.random.default_rng(21)
    n, V, L, B, c = 3000, 500, 24, 37, 20
    offs, terms = _postings(rng, n, V, L)
    # ragged term lists: drop a few terms of some rows
    m = rng.integers(0, 9, B)  # 0..8 query terms, including queries without keywords
    qoffs = np.concatenate([[0], np.cumsum(m)]).astype(np.int32)
    qterms = np.concatenate([rng.choice(V, mm, replace=False) for mm in m] + [np.empty(0, np.int64)]).astype(np.int32)
    rows = rng.integers(0, n, (B, c)).astype(np.int64)
    rows[5, 15:] = -1
    want = so.keyword_counts(offs, terms, qoffs, qterms, rows)
    post = eng.postings(torch.from_numpy(offs).cuda(), torch.from_numpy(terms).cuda())
    dq = torch.from_numpy(qterms).cuda() if qterms.size else torch.zeros(1, dtype=torch.int32, device="cuda")
    got = eng.keyword_counts(post, dq, torch.from_numpy(qoffs).cuda(), torch.from_numpy(rows).cuda())
    assert np.array_equal(got.cpu().numpy(), want)

    dist = np.sort(rng.uniform(0.0, 2.2, (B, c)).astype(np.float32), axis=1)
    dist[1, :] = 0.75  # all tied: stable order must keep the candidate order
    dist[2, 3:9] = dist[2, 3]
    dist[5, 15:] = np.inf
    nvalid = np.full(B, c, np.int32)
    nvalid[5] = 15
    for n_keep in (10, 20):
        wc, wp, _, _ = so.fuse_sort(dist, want, np.diff(qoffs), n_keep, 0.3, nvalid)
        gc, gp = eng.fuse_sort(torch.from_numpy(dist).cuda(), got, torch.from_numpy(qoffs).cuda(), n_keep, 0.3)
        assert np.array_equal(gp.cpu().numpy(), wp)
        assert np.array_equal(gc.cpu().numpy().view(np.uint64), wc.view(np.uint64))  # bit-exact fp64

    G, k = 4, 20
    keys = rng.integers(0, 2**62, (G, B, k)).astype(np.uint64)
    keys.sort(axis=2)
    keys[2, 7, 10:] = so.PAD_KEY
    vals = rng.integers(0, 9, (G, B, k)).astype(np.int32)
    wk = so.merge(keys)
    gk, gv = eng.merge_topk(torch.from_numpy(keys.view(np.int64)).cuda(), torch.from_numpy(vals).cuda())
    assert np.array_equal(_keys(gk), wk)
    lookup = {int(kk): int(v) for kk, v in zip(keys.reshape(-1), vals.reshape(-1))}
    gvn = gv.cpu().numpy()
    for b in range(B):
        for j in range(k):
            assert gvn[b, j] == lookup[int(wk[b, j])]
    post.close()


def test_keyword_counts_with_negative_query_terms_and_ragged_rows(eng):
    """Query terms may be any int32 (a term the vocabulary does not hold): lanes beyond the end of a row's term list
    must not compare as a match for any of them."""
    rng = np.random.default_rng(5)
    n, B, c = 400, 9, 12
    lens = rng.integers(0, 150, n)  # row lengths around and across the 64-term blocks, empty rows included
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    terms = np.concatenate([np.sort(rng.choice(1000, L, replace=False)) for L in lens]).astype(np.int32)
    qt = [np.array(t, np.int32) for t in ([-2], [-1, -2, 0], [5, -2, 7], [], [0], [-2147483648, 2147483647], [3, 3], [-2, -2], [1, 2, 3, 4])]
    qoffs = np.concatenate([[0], np.cumsum([len(t) for t in qt])]).astype(np.int32)
    qterms = np.concatenate(qt).astype(np.int32)
    rows = rng.integers(0, n, (B, c)).astype(np.int64)
    want = so.keyword_counts(offs, terms, qoffs, qterms, rows)
    assert want[0].max() == 0 and want[7].max() == 0
    post = eng.postings(torch.from_numpy(offs).cuda(), torch.from_numpy(terms).cuda())
    got = eng.keyword_counts(post, torch.from_numpy(qterms).cuda(), torch.from_numpy(qoffs).cuda(), torch.from_numpy(rows).cuda())
    assert np.array_equal(got.cpu().numpy(), want)
    post.close()


def test_keyword_counts_and_fusion_reproduce_the_reference(eng):
    """hrs_keyword_counts + hrs_fuse_sort against the outputs of the reference's own hybrid_query
    (scripts/chromadb_helper.py:296-325, 352-359; tests/golden/make_fusion_goldens.py): order, fused scores
    (fp64 bit patterns) and keyword scores, case by case."""
    import fusion_cases

    for c in fusion_cases.load():
        dev = lambda a: torch.from_numpy(a).cuda() if a.size else torch.zeros(1, dtype=torch.from_numpy(a).dtype, device="cuda")  # noqa: E731
        post = eng.postings(torch.from_numpy(c["doc_offs"]).cuda(), dev(c["term_ids"]))
        counts = eng.keyword_counts(post, dev(c["q_terms"]), torch.from_numpy(c["q_offs"]).cuda(), torch.from_numpy(c["rows"]).cuda())
        comb, perm = eng.fuse_sort(torch.from_numpy(c["dist"]).cuda(), counts, torch.from_numpy(c["q_offs"]).cuda(), c["n"], c["boost"])
        torch.cuda.synchronize()
        fusion_cases.check(c, counts.cpu().numpy(), comb.cpu().numpy(), perm.cpu().numpy())
        post.close()


def test_sharded_equals_unsharded(eng):
    """Row-shard the corpus 1/2/4/8 ways on one GPU, merge per-shard top-k, compare with one shard."""
    rng = np.random.default_rng(31)
    n, d, b, k = 9001, 384, 6, 20
    C = _unit(rng, n, d)
    Q = _unit(rng, b, d)
    want = so.topk_keys(Q, C, k, "l2")
    dq = torch.from_numpy(Q).cuda()
    for G in (1, 2, 4, 8):
        bounds = [n * g // G for g in range(G + 1)]
        parts, keep = [], []
        for g in range(G):
            shard = torch.from_numpy(C[bounds[g] : bounds[g + 1]]).cuda()
            cp = eng.corpus(shard, row_offset=bounds[g])
            keep.append(cp)
            parts.append(eng.score_topk(cp, dq, k, want_keys=True)[2])
        merged = eng.merge_topk(torch.stack(parts))
        torch.cuda.synchronize()
        assert np.array_equal(_keys(merged), want), G
        for cp in keep:
            cp.close()


def test_two_stream_pipeline_equals_single_stream(eng):
    """Alternate batches over two engines / streams (scans chained by hrs_ctx_set_peer): every batch
    gives what the plain single-stream step gives."""
    from hybrid_rag_system_amd.dist import HybridPipeline, ShardedHybrid
    from hybrid_rag_system_amd.engine import Engine

    rng = np.random.default_rng(41)
    n, d, b, keep, V, L = 80_000, 384, 96, 5, 500, 12
    C = _unit(rng, n, d)
    offs, terms = _postings(rng, n, V, L)
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    post = eng.postings(torch.from_numpy(offs).cuda(), torch.from_numpy(terms).cuda())
    m = rng.integers(1, 5, b)
    qoffs = torch.from_numpy(np.concatenate([[0], np.cumsum(m)]).astype(np.int32)).cuda()
    qterms = torch.from_numpy(rng.integers(0, V, int(m.sum())).astype(np.int32)).cuda()
    batches = [torch.from_numpy(_unit(rng, b, d)).cuda() for _ in range(5)]
    plain = ShardedHybrid(eng, corpus, post, group=False)
    want = [plain.step(q, qterms, qoffs, keep, mode="mfma") for q in batches]
    torch.cuda.synchronize()
    eng2 = Engine(0)
    pipe = HybridPipeline([eng, eng2], corpus, post, group=False)
    got = [r for r in (pipe.submit(q, qterms, qoffs, keep, mode="mfma") for q in batches) if r is not None]  # batch i comes back from submit i + 1
    got.append(pipe.drain())
    assert len(got) == len(batches)
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for key in ("cand_rows", "cand_dist", "cand_counts", "perm", "combined"):
            assert torch.equal(w[key], g[key]), key
    eng.set_peer(None)
    eng2.close()
    corpus.close()


def test_pipeline_results_are_consumable_on_the_callers_stream_without_a_device_sync(eng):
    """The stream contract of HybridPipeline: what submit returns may be read on the caller's current stream at
    once, and the inputs may be dropped right after submit -- no torch.cuda.synchronize() anywhere in the loop."""
    from hybrid_rag_system_amd.dist import HybridPipeline, ShardedHybrid
    from hybrid_rag_system_amd.engine import Engine

    rng = np.random.default_rng(43)
    n, d, b, keep, V, L = 400_000, 384, 512, 5, 500, 12
    C = _unit(rng, n, d)
    terms = np.sort(rng.integers(0, V, (n, L)).astype(np.int32), axis=1).reshape(-1)
    offs = np.arange(n + 1, dtype=np.int64) * L
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    post = eng.postings(torch.from_numpy(offs).cuda(), torch.from_numpy(terms).cuda())
    m = rng.integers(1, 5, b)
    qoffs_h = np.concatenate([[0], np.cumsum(m)]).astype(np.int32)
    qterms_h = rng.integers(0, V, int(m.sum())).astype(np.int32)
    host = [_unit(rng, b, d) for _ in range(6)]
    plain = ShardedHybrid(eng, corpus, post, group=False)
    want = []
    for q in host:
        o = plain.step(torch.from_numpy(q).cuda(), torch.from_numpy(qterms_h).cuda(), torch.from_numpy(qoffs_h).cuda(), keep, mode="mfma")
        torch.cuda.synchronize()
        want.append({k: v.cpu() for k, v in o.items()})
    eng2 = Engine(0)
    pipe = HybridPipeline([eng, eng2], corpus, post, group=False)
    side = torch.cuda.Stream()  # a caller that works on a stream of its own
    copies = []
    with torch.cuda.stream(side):
        for q in host:
            dq, dt, do = torch.from_numpy(q).cuda(), torch.from_numpy(qterms_h).cuda(), torch.from_numpy(qoffs_h).cuda()
            prev = pipe.submit(dq, dt, do, keep, mode="mfma")
            del dq, dt, do  # the allocator may hand these blocks out again: the pipeline has recorded its use of them
            junk = torch.full((b, d), 3.0, device="cuda")  # ... for instance to this
            if prev is not None:
                copies.append({k: v + 0 for k, v in prev.items()})  # read on the caller's stream, no sync
            del prev, junk
        last = pipe.drain()
        copies.append({k: v + 0 for k, v in last.items()})
    torch.cuda.synchronize()
    assert len(copies) == len(host)
    for w, g in zip(want, copies):
        for key in ("cand_rows", "cand_dist", "cand_counts", "perm", "combined"):
            assert torch.equal(w[key], g[key].cpu()), key
    eng.set_peer(None)
    eng2.close()
    corpus.close()
    post.close()


def test_two_phase_call_equals_single_call(eng):
    """hrs_score_topk_begin + _finish give what hrs_score_topk gives, whether the call can be split (matrix-core
    path, one chunk) or not (exact path, more than 1024 queries); one call may be open per context."""
    from hybrid_rag_system_amd._native import HrsError

    rng = np.random.default_rng(77)
    n, d = 70_000, 384
    corpus = eng.corpus(torch.from_numpy(_unit(rng, n, d)).cuda())
    mask = torch.from_numpy((rng.random(n) < 0.5).astype(np.uint8)).cuda()
    for b, mode, m in ((96, "mfma", None), (300, "auto", None), (3, "auto", None), (64, "mfma", mask), (1100, "mfma", None), (5, "exact", None)):
        q = torch.from_numpy(_unit(rng, b, d)).cuda()
        want = eng.score_topk(corpus, q, 10, mode=mode, mask=m, want_keys=True)
        eng.score_topk_begin(corpus, q, 10, mode=mode, mask=m)
        with pytest.raises(HrsError, match="not finished"):
            eng.score_topk_begin(corpus, q, 10, mode=mode, mask=m)
        got = eng.score_topk_finish(want_keys=True)
        torch.cuda.synchronize()
        for w, g in zip(want, got):
            assert torch.equal(w, g), (b, mode)
    with pytest.raises(ValueError):
        eng.score_topk_finish()
    corpus.close()


def test_hybrid_pipeline_vs_oracle(eng):
    rng = np.random.default_rng(41)
    n, d, B, n_keep, V, L = 120_000, 384, 260, 10, 2000, 32
    C = _unit(rng, n, d)
    Q = _unit(rng, B, d)
    for j in range(0, 60):
        Q[j] = C[j * 1999] + 0.3 * rng.standard_normal(d).astype(np.float32) / np.sqrt(d)
        Q[j] /= np.linalg.norm(Q[j])
    terms = np.sort(rng.integers(0, V, (n, L)).astype(np.int32), axis=1)  # duplicates allowed inside a row
    offs = np.arange(n + 1, dtype=np.int64) * L
    m = rng.integers(1, 9, B)
    qoffs = np.concatenate([[0], np.cumsum(m)]).astype(np.int32)
    qterms = np.concatenate([rng.choice(V, mm, replace=False) for mm in m]).astype(np.int32)
    # oracle
    wkeys = so.topk_keys(Q, C, 2 * n_keep, "l2")
    wd, wr = so.unpack(wkeys)
    wcnt = so.keyword_counts(offs, terms.reshape(-1), qoffs, qterms, wr)
    wc, wp, _, _ = so.fuse_sort(wd, wcnt, np.diff(qoffs), n_keep, 0.3)
    # GPU
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    post = eng.postings(torch.from_numpy(offs).cuda(), torch.from_numpy(terms.reshape(-1)).cuda())
    out = eng.hybrid_topk(corpus, post, torch.from_numpy(Q).cuda(), torch.from_numpy(qterms).cuda(), torch.from_numpy(qoffs).cuda(), n_keep, mode="mfma")
    torch.cuda.synchronize()
    assert np.array_equal(out["cand_rows"].cpu().numpy(), wr)
    assert np.array_equal(out["cand_dist"].cpu().numpy(), wd)
    assert np.array_equal(out["cand_counts"].cpu().numpy(), wcnt)
    assert np.array_equal(out["perm"].cpu().numpy(), wp)
    assert np.array_equal(out["combined"].cpu().numpy().view(np.uint64), wc.view(np.uint64))
    assert (wc[:60, 0] > 0.3).any()  # the clamp max(0, 1-d) is not always active
    corpus.close()
    post.close()


def test_synth_rows_shard_invariant(eng):
    a = eng.synth_rows(1234, 0, 1000, 384)
    b = eng.synth_rows(1234, 400, 300, 384)
    torch.cuda.synchronize()
    assert torch.equal(a[400:700], b)
    nrm = a.norm(dim=1)
    assert (nrm - 1).abs().max() < 1e-5
    assert abs(float(a.mean())) < 1e-3 and abs(float(a.std()) - 1 / np.sqrt(384)) < 1e-3
    cdf = torch.cumsum(1.0 / torch.arange(1, 5001, dtype=torch.float64), 0)
    cdf = (cdf / cdf[-1]).float().cuda()
    p = eng.synth_postings(777, 0, 2000, cdf, 96)
    p2 = eng.synth_postings(777, 1000, 500, cdf, 96)
    torch.cuda.synchronize()
    assert torch.equal(p[1000:1500], p2)
    pn = p.cpu().numpy()
    assert (np.diff(pn, axis=1) > 0).all() and pn.min() >= 0 and pn.max() < 5000
    assert (pn == 0).any(axis=1).mean() > 0.5  # the most popular term is in most rows (Zipf)


@pytest.mark.parametrize("shadow,d", [("i8", 384), ("bf16", 384), ("bf16", 1024), ("none", 384)])
def test_corpus_append_equals_fresh_registration(eng, shadow, d):
    """hrs_corpus_append: rows arrive in ragged batches (partly filled 32-row groups are re-quantised, the shadow's
    allocation doubles twice); after every append the matrix-core path and the exact scan return the keys of a
    corpus registered in one go -- the oracle's."""
    rng = np.random.default_rng(17 + d)
    sizes = [70_001, 13, 19, 1, 40_000, 95_000, 31]
    total = sum(sizes)
    C = _unit(rng, total, d)
    C[total - 5] = C[3]
    Q = _unit(rng, 48, d)
    Q[0] = C[3]
    buf = torch.empty((total, d), dtype=torch.float32, device="cuda")
    n = sizes[0]
    buf[:n] = torch.from_numpy(C[:n]).cuda()
    corpus = eng.corpus(buf, shadow=shadow, n=n)
    dq = torch.from_numpy(Q).cuda()
    for step in sizes[1:]:
        buf[n : n + step] = torch.from_numpy(C[n : n + step]).cuda()
        n += step
        corpus.append(buf, n)
        if step in (1, 95_000, 31):
            want = so.topk_keys(Q, C[:n], 10, "l2")
            for mode in ("mfma", "exact"):
                _, _, keys = eng.score_topk(corpus, dq, 10, mode=mode, want_keys=True)
                torch.cuda.synchronize()
                assert np.array_equal(_keys(keys), want), (n, mode, eng.score_stats())
    mn, mx = corpus.norm_stats()
    nn = (C.astype(np.float64) ** 2).sum(1)
    assert abs(mn - nn.min()) < 1e-9 and abs(mx - nn.max()) < 1e-9
    # moved to a new address (the caller's buffer was reallocated) and grown from empty
    moved = torch.from_numpy(C).cuda()
    corpus.append(moved, total)
    _, _, keys = eng.score_topk(corpus, dq, 10, mode="mfma", want_keys=True)
    assert np.array_equal(_keys(keys), so.topk_keys(Q, C, 10, "l2"))
    corpus.close()
    empty = eng.corpus(moved, n=0)
    empty.append(moved, 5000)
    _, _, keys = eng.score_topk(empty, dq, 10, want_keys=True)
    torch.cuda.synchronize()
    assert np.array_equal(_keys(keys), so.topk_keys(Q, C[:5000], 10, "l2"))
    empty.close()


def test_norm_stats_without_a_shadow(eng):
    rng = np.random.default_rng(3)
    C = (rng.standard_normal((3000, 1500)) * rng.uniform(0.1, 3, (3000, 1))).astype(np.float32)  # deeper than any shadow
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    mn, mx = corpus.norm_stats()
    nn = (C.astype(np.float64) ** 2).sum(1)
    assert abs(mn / nn.min() - 1) < 1e-12 and abs(mx / nn.max() - 1) < 1e-12
    corpus.close()


def test_topk_resumed_behind_a_key_concatenates_to_the_exact_order(eng):
    """hrs_score_topk_after: three passes of 128 resumed behind each query's last key equal one top-384 of the oracle,
    including through a run of exact ties that straddles a pass boundary and a query that runs out of rows."""
    rng = np.random.default_rng(23)
    n, d, b = 5000, 64, 6
    C = _unit(rng, n, d)
    C[100:400] = C[7]  # 301 exact ties: the boundary of pass 1 falls inside the run
    Q = _unit(rng, b, d)
    Q[0] = C[7]
    mask = np.ones(n, np.uint8)
    mask[rng.random(n) < 0.5] = 0
    want = so.topk_keys(Q, C, 384, "l2")
    wantm = so.topk_keys(Q, C, 384, "l2", mask=mask)
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    dq = torch.from_numpy(Q).cuda()
    for m, w in ((None, want), (torch.from_numpy(mask).cuda(), wantm)):
        _, _, keys = eng.score_topk(corpus, dq, 128, mask=m, want_keys=True)
        got = [keys]
        for _ in range(2):
            _, _, keys = eng.score_topk_after(corpus, dq, 128, got[-1][:, -1].contiguous(), mask=m)
            got.append(keys)
        torch.cuda.synchronize()
        assert np.array_equal(_keys(torch.cat(got, dim=1)), w)
    small = eng.corpus(torch.from_numpy(C[:150]).cuda())
    _, r1, k1 = eng.score_topk(small, dq, 128, want_keys=True)
    _, r2, k2 = eng.score_topk_after(small, dq, 128, k1[:, -1].contiguous())
    _, r3, _ = eng.score_topk_after(small, dq, 128, k2[:, -1].contiguous())
    torch.cuda.synchronize()
    assert (r2[:, :22] >= 0).all() and (r2[:, 22:] == -1).all() and (r3 == -1).all()
    small.close()
    corpus.close()


@pytest.mark.parametrize("space", ["l2", "cosine"])
def test_full_corpus_hybrid_equals_brute_force(eng, space):
    """The full-corpus hybrid mode (an extension behind its own entry points; the reference's re-rank of the 2n
    nearest stays the default): match counts of EVERY row from inverted postings, the best n of every match-count
    class, fusion -- against the oracle's brute force over all rows (own semantics, parity unpinned)."""
    from hybrid_rag_system_amd.engine import InvertedPostings

    rng = np.random.default_rng(61)
    n, d, B, V, n_keep = 30_011, 96, 23, 300, 10
    C = _unit(rng, n, d)
    C[200:260] = C[7]  # ties inside and across match-count classes
    Q = _unit(rng, B, d)
    for j in range(8):
        Q[j] = C[j * 1000 + 7] + 0.2 * rng.standard_normal(d).astype(np.float32) / np.sqrt(d)
        Q[j] /= np.linalg.norm(Q[j])
    # ragged term lists (0..12 distinct terms per row, Zipf-like popularity so that some terms are in most rows)
    p = 1.0 / np.arange(1, V + 1)
    p /= p.sum()
    lens = rng.integers(0, 13, n)
    lists = [np.sort(rng.choice(V, int(l), replace=False, p=p)).astype(np.int32) for l in lens]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    terms = np.concatenate(lists + [np.empty(0, np.int32)]).astype(np.int32)
    m = rng.integers(0, 8, B)  # queries without keywords too
    m[0] = 7
    qoffs = np.concatenate([[0], np.cumsum(m)]).astype(np.int32)
    qterms = np.concatenate([rng.choice(V, int(mm), replace=False, p=p) for mm in m] + [np.empty(0, np.int64)]).astype(np.int32)
    qterms[0] = 0  # the most popular term
    if m[0] > 1:
        qterms[1] = V - 1  # and a rare one
    want_rows, want_comb, want_cnt, want_dist = so.hybrid_full(Q, C, offs, terms, qoffs, qterms, n_keep, 0.3, space)

    corpus = eng.corpus(torch.from_numpy(C).cuda())
    inv = InvertedPostings.from_forward(eng, torch.from_numpy(offs).cuda(), torch.from_numpy(terms).cuda(), V)
    # the match counts of every row equal a host count over the forward lists
    counts = eng.match_counts(inv, torch.from_numpy(qterms).cuda(), torch.from_numpy(qoffs).cuda(), B)
    torch.cuda.synchronize()
    hc = np.zeros((B, n), np.uint8)
    for b in range(B):
        qs = set(qterms[qoffs[b] : qoffs[b + 1]].tolist())
        hc[b] = [len(qs.intersection(l.tolist())) for l in lists]
    assert np.array_equal(counts.cpu().numpy()[:, :n], hc)
    for chunk in (4 << 30, 5 * n):  # all queries at once, and in chunks of five
        out = eng.hybrid_full_corpus(corpus, inv, torch.from_numpy(Q).cuda(), torch.from_numpy(qterms).cuda(), torch.from_numpy(qoffs).cuda(), n_keep, 0.3, space, chunk_bytes=chunk)
        torch.cuda.synchronize()
        assert np.array_equal(out["rows"].cpu().numpy(), want_rows)
        assert np.array_equal(out["counts"].cpu().numpy(), want_cnt)
        assert np.array_equal(out["dist"].cpu().numpy(), want_dist)
        assert np.array_equal(out["combined"].cpu().numpy().view(np.uint64), want_comb.view(np.uint64))
    # row-sharded: every shard's nearest n of every class, merged per class (what dist.exchange_classes does after its
    # all-gather), then the same second half -- equal to the unsharded result
    qd, td, od = torch.from_numpy(Q).cuda(), torch.from_numpy(qterms).cuda(), torch.from_numpy(qoffs).cuda()
    levels = int(m.max()) + 1
    for G in (2, 3):
        bounds = [n * g // G for g in range(G + 1)]
        parts, keep = [], []
        for g in range(G):
            lo, hi = bounds[g], bounds[g + 1]
            cp = eng.corpus(torch.from_numpy(C[lo:hi]).cuda(), row_offset=lo)
            so_, st_ = torch.from_numpy(offs[lo : hi + 1] - offs[lo]).cuda(), torch.from_numpy(terms[offs[lo] : offs[hi]]).cuda()
            iv = InvertedPostings.from_forward(eng, so_, st_ if st_.numel() else torch.zeros(1, dtype=torch.int32, device="cuda"), V)
            keep += [cp, iv]
            parts.append(eng.full_corpus_local(cp, iv, qd, td, od, n_keep, levels, space))
        merged = eng.merge_classes(torch.stack(parts))
        got = eng.full_corpus_finish(merged, od, n_keep, 0.3)
        torch.cuda.synchronize()
        assert np.array_equal(got["rows"].cpu().numpy(), want_rows), G
        assert np.array_equal(got["combined"].cpu().numpy().view(np.uint64), want_comb.view(np.uint64)), G
        for h in keep:
            h.close()
    # it differs from the reference's re-rank where a keyword match lies outside the 2n nearest rows
    post = eng.postings(torch.from_numpy(offs).cuda(), torch.from_numpy(terms).cuda())
    rr = eng.hybrid_topk(corpus, post, torch.from_numpy(Q).cuda(), torch.from_numpy(qterms).cuda(), torch.from_numpy(qoffs).cuda(), n_keep, space=space)
    torch.cuda.synchronize()
    assert (rr["combined"].cpu().numpy()[:, 0] <= want_comb[:, 0] + 1e-15).all() and (rr["combined"].cpu().numpy() < want_comb - 1e-9).any()
    inv.close()
    post.close()
    corpus.close()


@pytest.mark.parametrize("space,n,d,b,k,levels", [("l2", 300_037, 32, 6, 10, 9), ("cosine", 150_001, 100, 5, 100, 7), ("ip", 70_000, 384, 3, 128, 3), ("l2", 1000, 64, 1, 5, 200)])
def test_topk_by_classes_in_one_pass(eng, space, n, d, b, k, levels):
    """hrs_score_topk_levels: the nearest k rows of every class of every query, all classes of four queries in one
    pass over the corpus (groups of 256 / k classes per pass where they do not all fit: k = 100 and 128 here; several
    tiles per workgroup so that lists are compacted on the way; a class byte beyond `levels` belongs to no class),
    against one masked oracle search per (query, class)."""
    rng = np.random.default_rng(n + k)
    C = _unit(rng, n, d)
    if space != "l2":
        C *= rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
    C[n // 3 : n // 3 + 40] = C[5]  # ties within and across classes
    Q = _unit(rng, b, d)
    Q[0] = C[5]
    pitch = (n + 3) // 4 * 4
    cls = np.zeros((b, pitch), np.uint8)
    # skewed classes: most rows in class 0, the highest classes nearly or entirely empty
    cls[:, :n] = np.minimum(rng.geometric(0.5, (b, n)) - 1, levels + 1).astype(np.uint8)
    if levels > 2:
        cls[0, :n][cls[0, :n] == levels - 1] = 0  # an empty class
    off = 5000
    corpus = eng.corpus(torch.from_numpy(C).cuda(), row_offset=off)
    keys = eng.score_topk_levels(corpus, torch.from_numpy(Q).cuda(), k, levels, torch.from_numpy(cls).cuda(), space=space)
    torch.cuda.synchronize()
    got = _keys(keys)
    assert got.shape == (b, levels, k)
    for q in range(b):
        for l in range(levels):
            want = so.topk_keys(Q[q : q + 1], C, k, space, row_offset=off, mask=(cls[q, :n] == l).astype(np.uint8))
            assert np.array_equal(got[q, l], want[0]), (q, l)
    corpus.close()


def test_bm25_keyword_scores_and_fusion(eng):
    """The BM25 keyword weighting (an extension with its own semantics: parity unpinned): hrs_bm25_scores and
    hrs_fuse_sort_scores against the oracle's Python-float restatement, bit for bit -- ragged term lists with repeated
    terms, empty rows, queries without terms, terms outside the vocabulary, rows outside the shard, missing candidates."""
    rng = np.random.default_rng(77)
    n, V, B, c = 5000, 400, 37, 20
    p = 1.0 / np.arange(1, V + 1)
    p /= p.sum()
    lens = rng.integers(0, 150, n)
    lens[:5] = 0
    lists = [rng.choice(V, int(l), replace=True, p=p).astype(np.int32) for l in lens]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    terms = np.concatenate(lists + [np.empty(0, np.int32)]).astype(np.int32)
    df = np.zeros(V, np.int64)
    for l in lists:
        df[np.unique(l)] += 1
    idf = so.bm25_idf(df, n)
    avg_len = float(lens.mean())
    m = rng.integers(0, 9, B)
    m[1] = 70  # more terms than a wave has lanes
    qoffs = np.concatenate([[0], np.cumsum(m)]).astype(np.int32)
    qterms = np.concatenate([rng.choice(V, int(mm), replace=True, p=p) for mm in m] + [np.empty(0, np.int64)]).astype(np.int32)
    qterms[qoffs[2]] = -7       # ignored
    qterms[qoffs[3]] = V + 3    # ignored
    off = 1000
    rows = rng.integers(off, off + n, (B, c)).astype(np.int64)
    rows[0, :5] = off + np.arange(5)  # empty rows
    rows[4, 3] = -1
    rows[4, 4] = off - 1
    rows[4, 5] = off + n
    want = so.bm25_scores(offs, terms, idf, avg_len, qoffs, qterms, rows, row_offset=off)
    post = eng.postings(torch.from_numpy(offs).cuda(), torch.from_numpy(terms).cuda(), row_offset=off)
    got = eng.bm25_scores(post, torch.from_numpy(idf).cuda(), avg_len, torch.from_numpy(qterms).cuda(), torch.from_numpy(qoffs).cuda(), torch.from_numpy(rows).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), want.view(np.uint64))
    assert (want >= 0).all() and (want < 1).all() and (want > 0).any()
    dist = np.sort(rng.random((B, c)).astype(np.float32) * 1.3, axis=1)
    dist[4, 15:] = np.inf
    dist[7, 3] = dist[7, 2]
    wc, wp = so.fuse_sort_scores(dist, want, 10, 0.3)
    gc, gp = eng.fuse_sort_scores(torch.from_numpy(dist).cuda(), got, 10, 0.3)
    torch.cuda.synchronize()
    assert np.array_equal(gp.cpu().numpy(), wp)
    assert np.array_equal(gc.cpu().numpy().view(np.uint64), wc.view(np.uint64))
    post.close()


def test_sharded_bm25_step_equals_unsharded(eng):
    """BM25 weights on a row-sharded corpus: every shard scores its own top-2n with the statistics of the whole corpus,
    the fp64 scores pass through hrs_merge_topk_i64 beside their keys as 64 opaque bits, and the fused result equals
    the unsharded Engine.hybrid_topk(bm25=...) and the oracle, bit for bit (2, 3 and 8 shards)."""
    rng = np.random.default_rng(912)
    n, d, b, keep, V = 20_003, 96, 33, 6, 300
    C = _unit(rng, n, d)
    p = 1.0 / np.arange(1, V + 1)
    p /= p.sum()
    lens = rng.integers(1, 40, n)
    lists = [rng.choice(V, int(l), replace=True, p=p).astype(np.int32) for l in lens]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    terms = np.concatenate(lists).astype(np.int32)
    df = np.zeros(V, np.int64)
    for l in lists:
        df[np.unique(l)] += 1
    idf = so.bm25_idf(df, n)
    avg_len = float(lens.mean())
    m = rng.integers(0, 6, b)
    qoffs = np.concatenate([[0], np.cumsum(m)]).astype(np.int32)
    qterms = rng.choice(V, int(m.sum()), replace=True, p=p).astype(np.int32)
    Q = _unit(rng, b, d)
    dq, dqt, dqo = torch.from_numpy(Q).cuda(), torch.from_numpy(qterms).cuda(), torch.from_numpy(qoffs).cuda()
    bm25 = {"idf": torch.from_numpy(idf).cuda(), "avg_len": avg_len}
    # the oracle's answer
    wd, wr = so.unpack(so.topk_keys(Q, C, 2 * keep, "l2"))
    wkw = so.bm25_scores(offs, terms, idf, avg_len, qoffs, qterms, wr)
    wc, wp = so.fuse_sort_scores(wd, wkw, keep, 0.3)
    assert (wkw > 0).any()
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    post = eng.postings(torch.from_numpy(offs).cuda(), torch.from_numpy(terms).cuda())
    one = eng.hybrid_topk(corpus, post, dq, dqt, dqo, keep, bm25=bm25)
    torch.cuda.synchronize()
    assert np.array_equal(one["cand_rows"].cpu().numpy(), wr) and np.array_equal(one["perm"].cpu().numpy(), wp)
    assert np.array_equal(one["combined"].cpu().numpy().view(np.uint64), wc.view(np.uint64))
    corpus.close()
    post.close()
    for G in (2, 3, 8):
        bounds = [n * g // G for g in range(G + 1)]
        keys, kws, held = [], [], []
        for g in range(G):
            lo, hi = bounds[g], bounds[g + 1]
            cp = eng.corpus(torch.from_numpy(C[lo:hi]).cuda(), row_offset=lo)
            ps = eng.postings(torch.from_numpy(offs[lo : hi + 1] - offs[lo]).cuda(), torch.from_numpy(terms[offs[lo] : offs[hi]]).cuda(), row_offset=lo)
            held += [cp, ps]
            _, rows, k64 = eng.score_topk(cp, dq, 2 * keep, want_keys=True)
            keys.append(k64)
            kws.append(eng.bm25_scores(ps, bm25["idf"], avg_len, dqt, dqo, rows))
        mk, mkw = eng.merge_topk(torch.stack(keys), torch.stack(kws))
        md, mr = eng.unpack_keys(mk)
        gc, gp = eng.fuse_sort_scores(md, mkw, keep, 0.3)
        torch.cuda.synchronize()
        assert mkw.dtype == torch.float64
        assert np.array_equal(mr.cpu().numpy(), wr), G
        assert np.array_equal(mkw.cpu().numpy().view(np.uint64), wkw.view(np.uint64)), G
        assert np.array_equal(gp.cpu().numpy(), wp) and np.array_equal(gc.cpu().numpy().view(np.uint64), wc.view(np.uint64)), G
        for h in held:
            h.close()


def test_main_pass_share_is_reported(eng):
    """hrs_scan_work: the second sampled pass records the rows of its own tiles (one per block of its stride) and the
    main pass of the large int8 batches leaves those tiles out; the library reports what share of the corpus' tiles
    that main pass scored itself, and the keys are the oracle's all the same (a tile left out twice, or scored
    twice, would show as a missing or a doubled key)."""
    rng = np.random.default_rng(404)
    n, d, b, k = 400_003, 128, 600, 20
    C = _unit(rng, n, d)
    Q = _unit(rng, b, d)
    Q[:50] = C[rng.integers(0, n, 50)] + 0.05 * rng.standard_normal((50, d)).astype(np.float32)
    corpus = eng.corpus(torch.from_numpy(C).cuda())
    _, _, keys = eng.score_topk(corpus, torch.from_numpy(Q).cuda(), k, mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    share = eng.scan_work()
    assert 0.9 < share < 1.0, share
    want = so.topk_keys(Q, C, k, "l2")
    assert np.array_equal(_keys(keys), want)
    st = eng.score_stats()
    assert st["fallback_queries"] <= 6, st
    corpus.close()
