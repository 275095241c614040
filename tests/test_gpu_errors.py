"""Error behaviour of the C ABI: a bad call returns a negative code, leaves a message for hrs_last_error,
launches nothing, and the context keeps working afterwards."""
import ctypes as C

import numpy as np
import pytest
import torch

from hybrid_rag_system_amd import _native as N
from hybrid_rag_system_amd.engine import Engine
from oracle import search_oracle as so

pytestmark = pytest.mark.gpu
ERR_ARG = -1


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def _p(t):
    return C.c_void_p(t.data_ptr())


def test_bad_arguments_are_rejected_with_a_message(eng):
    L = N.lib()
    rows = torch.randn(4096, 64, device="cuda")
    q = torch.randn(3, 64, device="cuda")
    keys = torch.zeros(3 * 200, dtype=torch.int64, device="cuda")
    h = C.c_void_p()
    # corpus shapes
    for n, d in ((16, 6), (16, 0), (16, 4096), (-1, 64), (1 << 33, 64)):
        assert L.hrs_corpus_register(eng._ctx, _p(rows), n, d, 0, None, C.byref(h)) == ERR_ARG
        assert "hrs_corpus_register" in N.last_error()
    assert L.hrs_corpus_register(eng._ctx, None, 16, 64, 0, None, C.byref(h)) == ERR_ARG
    assert L.hrs_corpus_register(eng._ctx, _p(rows), 16, 64, 0, None, None) == ERR_ARG
    assert L.hrs_corpus_register(None, _p(rows), 16, 64, 0, None, C.byref(h)) == ERR_ARG
    # a good one
    N.check(L.hrs_corpus_register(eng._ctx, _p(rows), 4096, 64, 0, None, C.byref(h)), "register")
    call = lambda B, k, space, mode, qp=_p(q), ctx=eng._ctx, corp=h: L.hrs_score_topk(ctx, corp, qp, B, k, space, None, mode, _p(keys), None, None, None)
    for args, word in (((3, 0, 0, 0), "k=0"), ((3, N.MAX_K + 1, 0, 0), "k="), ((3, 5, 3, 0), "space"), ((3, 5, -1, 0), "space"), ((3, 5, 0, 7), "mode"), ((-1, 5, 0, 0), "B=")):
        assert call(*args) == ERR_ARG, args
        assert word in N.last_error(), (args, N.last_error())
    assert call(3, 5, 0, 0, qp=None) == ERR_ARG and "null queries" in N.last_error()
    assert call(3, 5, 0, 0, ctx=None) == ERR_ARG
    assert call(3, 5, 0, 0, corp=None) == ERR_ARG
    assert L.hrs_corpus_shadow(eng._ctx, h, 9, None) == ERR_ARG and "kind" in N.last_error()
    assert L.hrs_ctx_set_peer(None, eng._ctx) == ERR_ARG
    assert L.hrs_unpack_keys(eng._ctx, None, 4, None, None, None) == ERR_ARG
    out = C.c_void_p()
    assert L.hrs_ctx_create(99, C.byref(out)) == ERR_ARG and "out of range" in N.last_error()
    assert L.hrs_ctx_create(0, None) == ERR_ARG
    # B = 0 is a valid empty call
    assert call(0, 5, 0, 0, qp=None) == 0
    # the context still answers correctly after all of that
    torch.cuda.synchronize()
    keys.fill_(-1)
    assert call(3, 5, 0, 0) == 0
    torch.cuda.synchronize()
    d, r = eng.unpack_keys(keys[:15].view(3, 5))
    wd, wr = so.topk(q.cpu().numpy(), rows.cpu().numpy(), 5, "l2")
    assert np.array_equal(r.cpu().numpy(), wr) and np.array_equal(d.cpu().numpy(), wd)
    assert L.hrs_corpus_release(eng._ctx, h) == 0
    # releasing nothing is harmless
    assert L.hrs_corpus_release(eng._ctx, None) == 0 and L.hrs_ctx_destroy(None) == 0


def test_wrapper_raises_hrs_error(eng):
    corpus = eng.corpus(torch.randn(256, 32, device="cuda"))
    q = torch.randn(2, 32, device="cuda")
    with pytest.raises(N.HrsError, match="k=0"):
        eng.score_topk(corpus, q, 0)
    with pytest.raises(N.HrsError):
        eng.corpus(torch.randn(8, 6, device="cuda"))
    with pytest.raises(KeyError):
        eng.score_topk(corpus, q, 3, space="manhattan")
    d, r = eng.score_topk(corpus, q, 3)
    assert r.shape == (2, 3)
    corpus.close()


def test_bad_arguments_of_the_rerank_entry_points(eng):
    """The fusion / re-rank entry points (keyword classes, BM25 scores, fusion with given scores, the merge with 64-bit
    payloads) reject sizes their kernels are not built for before anything is launched."""
    L = N.lib()
    f8 = C.c_double
    keys = torch.zeros(4 * 300, dtype=torch.int64, device="cuda")
    i32 = torch.zeros(4 * 300, dtype=torch.int32, device="cuda")
    f32 = torch.zeros(4 * 300, dtype=torch.float32, device="cuda")
    f64 = torch.zeros(4 * 300, dtype=torch.float64, device="cuda")
    offs = torch.zeros(5, dtype=torch.int32, device="cuda")
    # levels * k beyond one block of 256 candidates
    assert L.hrs_classes_finish(eng._ctx, _p(keys), _p(offs), 4, 9, 30, f8(0.3), 5, _p(keys), _p(f32), _p(i32), _p(f64), None) == ERR_ARG
    assert "levels * k" in N.last_error()
    assert L.hrs_classes_finish(eng._ctx, None, _p(offs), 4, 2, 3, f8(0.3), 5, _p(keys), _p(f32), _p(i32), _p(f64), None) == ERR_ARG
    assert L.hrs_classes_finish(eng._ctx, _p(keys), _p(offs), 4, 2, 3, f8(0.3), 0, _p(keys), _p(f32), _p(i32), _p(f64), None) == ERR_ARG
    # fusion with given keyword scores: at most 256 candidates, at least one result
    assert L.hrs_fuse_sort_scores(eng._ctx, _p(f32), _p(f64), 4, 257, f8(0.3), 5, _p(f64), _p(i32), None) == ERR_ARG
    assert "c=257" in N.last_error()
    assert L.hrs_fuse_sort_scores(eng._ctx, _p(f32), None, 4, 20, f8(0.3), 5, _p(f64), _p(i32), None) == ERR_ARG
    # the merge with 64-bit payloads: payload and both outputs are required, G * k <= 2048
    assert L.hrs_merge_topk_i64(eng._ctx, _p(keys), None, 2, 4, 10, _p(keys), _p(keys), None) == ERR_ARG
    assert L.hrs_merge_topk_i64(eng._ctx, _p(keys), _p(keys), 300, 4, 10, _p(keys), _p(keys), None) == ERR_ARG
    assert "G*k" in N.last_error()
    # BM25: statistics that make no sense
    terms = torch.zeros(8, dtype=torch.int32, device="cuda")
    doc_offs = torch.arange(0, 9, 2, dtype=torch.int64, device="cuda")
    post = eng.postings(doc_offs, terms)
    rows = torch.zeros((4, 3), dtype=torch.int64, device="cuda")
    bm = lambda avg, k1, b, idf=_p(f64): L.hrs_bm25_scores(eng._ctx, post._h, idf, 10, f8(avg), f8(k1), f8(b), _p(terms), _p(offs), _p(rows), 4, 3, _p(f64), None)
    assert bm(0.0, 1.2, 0.75) == ERR_ARG and "avg_len" in N.last_error()
    assert bm(2.0, -1.0, 0.75) == ERR_ARG
    assert bm(2.0, 1.2, 1.5) == ERR_ARG
    assert bm(2.0, 1.2, 0.75, idf=None) == ERR_ARG
    assert bm(2.0, 1.2, 0.75) == 0  # and a good call still goes through
    torch.cuda.synchronize()
    post.close()
