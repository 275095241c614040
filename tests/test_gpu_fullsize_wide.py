"""One GPU's shard of the 100M x 1024 configuration at full size (12.5M x 1024 fp32 = 51 GB, batch 1024,
cosine top-100) through properties that need no CPU oracle: the matrix-core path and the fp64 streaming scan
agree bit for bit on a sample, keys are strictly ascending, planted rows come first, a 2-way row shard +
hrs_merge_topk reproduces the unsharded answer, the exact fallback stays the exception."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, D, B, K = 12_500_000, 1024, 1024, 100


@pytest.fixture(scope="module")
def world():
    from hybrid_rag_system_amd.engine import Engine

    torch.cuda.empty_cache()
    eng = Engine(0)
    m = torch.empty((N, D), dtype=torch.float32, device="cuda")
    for r0 in range(0, N, 1 << 20):
        nr = min(1 << 20, N - r0)
        eng.synth_rows(1236, r0, nr, D, out=m[r0 : r0 + nr])
    q = eng.synth_rows(4321, 0, B, D)
    planted = torch.arange(0, 64, device="cuda") * 190_001 + 11
    q[torch.arange(0, 64, device="cuda") * 16] = m[planted]  # one planted query in every group of 16
    corpus = eng.corpus(m)
    yield eng, m, corpus, q.contiguous(), planted
    corpus.close()
    eng.close()
    del m
    torch.cuda.empty_cache()


def test_shard_of_100m_paths_agree_and_are_ordered(world):
    eng, m, corpus, q, planted = world
    d1, r1, k1 = eng.score_topk(corpus, q, K, space="cosine", mode="mfma", want_keys=True)
    torch.cuda.synchronize()
    st = eng.score_stats()
    assert st["overflow_queries"] == 0 and st["fallback_queries"] <= 8, st
    keys = k1.cpu().numpy().view(np.uint64)
    assert (keys[:, 1:] > keys[:, :-1]).all()  # strictly ascending: ordered, no row twice
    pq = torch.arange(0, 64, device="cuda") * 16
    assert (r1[pq, 0] == planted).all() and (d1[pq, 0].abs() < 1e-6).all()
    # the fp64 streaming scan gives the same keys (a sample: it serves one 1024-deep query per pass)
    sel = torch.tensor([0, 1, 16, 511, 512, 1023], device="cuda")
    _, _, k2 = eng.score_topk(corpus, q[sel].contiguous(), K, space="cosine", mode="exact", want_keys=True)
    assert torch.equal(k1[sel], k2)
    # idempotence
    _, _, k3 = eng.score_topk(corpus, q, K, space="cosine", mode="mfma", want_keys=True)
    assert torch.equal(k1, k3)


def test_shard_of_100m_split_in_two_merges_to_the_same(world):
    eng, m, corpus, q, planted = world
    _, _, want = eng.score_topk(corpus, q, K, space="cosine", mode="mfma", want_keys=True)
    parts, keep = [], []
    G = 2
    for g in range(G):
        lo, hi = N * g // G, N * (g + 1) // G
        cp = eng.corpus(m[lo:hi], row_offset=lo)
        keep.append(cp)
        parts.append(eng.score_topk(cp, q, K, space="cosine", mode="mfma", want_keys=True)[2])
    merged = eng.merge_topk(torch.stack(parts))
    torch.cuda.synchronize()
    assert torch.equal(merged, want)
    for cp in keep:
        cp.close()


def test_shard_of_100m_bf16_shadow_gives_the_same_keys(world):
    """The shadow is a choice of speed, never of results: the depth-1024 bf16 kernel (the default at this depth before
    the int8 bound was tightened) returns the keys of the int8 scan, at full size.  (Runs last: it rebuilds the shadow.)"""
    eng, m, corpus, q, planted = world
    _, _, want = eng.score_topk(corpus, q, K, space="cosine", mode="mfma", want_keys=True)
    corpus.set_shadow("bf16")
    try:
        _, _, got = eng.score_topk(corpus, q, K, space="cosine", mode="mfma", want_keys=True)
        torch.cuda.synchronize()
        st = eng.score_stats()
        assert st["overflow_queries"] == 0 and st["fallback_queries"] <= 8, st
        assert torch.equal(got, want)
    finally:
        corpus.set_shadow("auto")
