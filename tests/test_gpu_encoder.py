"""GPU encoder (bf16 MFMA) against the fp32 CPU oracle; tolerance from BASELINE.json: cosine >= 0.999."""
import numpy as np
import pytest
import torch

from hybrid_rag_system_amd import weights as W
from oracle import encoder_oracle as EO

pytestmark = pytest.mark.gpu

COS_TOL = 0.999  # north_star: "embeddings match within cosine >= 0.999"


@pytest.fixture(scope="module")
def setup():
    from hybrid_rag_system_amd.engine import Engine

    eng = Engine(0)
    cfg = W.EncoderConfig()
    w = W.generate(cfg, seed=2026)
    enc = eng.encoder(w, cfg)
    yield eng, enc, w, cfg
    enc.close()
    eng.close()


def _inputs(B, S, seed, full=False):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, 30000, (B, S)).astype(np.int32)
    lens = np.full(B, S, np.int32) if full else rng.integers(3, S + 1, B).astype(np.int32)
    lens[0] = S
    if B > 1 and not full:
        lens[1] = 1
    ids[:, 0] = 101
    for b in range(B):
        ids[b, lens[b] - 1] = 102
        ids[b, lens[b]:] = 0
    return ids, lens


@pytest.mark.parametrize("B,S", [(16, 256), (5, 64), (8, 32), (3, 160), (40, 128), (7, 96), (9, 192), (4, 224)])  # every sequence length the attention kernel is instantiated for
def test_encoder_matches_oracle(setup, B, S):
    eng, enc, w, cfg = setup
    ids, lens = _inputs(B, S, seed=B * 1000 + S)
    want = EO.forward(w, ids, lens, cfg).numpy()
    got = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-5
    cos = (got * want).sum(1)
    assert cos.min() >= COS_TOL, cos
    # random-token inputs share a large common component; the part that distinguishes inputs must match too
    if B >= 5:
        gc, wc = got - got.mean(0), want - want.mean(0)
        ccos = (gc * wc).sum(1) / (np.linalg.norm(gc, axis=1) * np.linalg.norm(wc, axis=1))
        assert ccos.min() > 0.98, ccos
    assert np.abs(got - want).max() < 5e-3


def test_padding_does_not_leak(setup):
    """Tokens past `lens` must not influence the embedding."""
    eng, enc, w, cfg = setup
    ids, lens = _inputs(8, 128, seed=5)
    a = enc.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
    ids2 = ids.copy()
    rng = np.random.default_rng(6)
    for b in range(8):
        ids2[b, lens[b]:] = rng.integers(1000, 30000, 128 - lens[b])
    b2 = enc.encode_ids(torch.from_numpy(ids2).cuda(), torch.from_numpy(lens).cuda()).cpu().numpy()
    assert np.array_equal(a, b2)


def test_full_batch_is_batch_invariant(setup):
    """Config 3's shape (1024 sequences x 256 tokens): unit rows, and the same rows whatever the batch
    they travel in (a sequence's result depends on nothing but its own tokens), bit for bit."""
    eng, enc, w, cfg = setup
    ids, lens = _inputs(1024, 256, seed=77)
    dids, dlens = torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()
    full = enc.encode_ids(dids, dlens)
    norms = full.norm(dim=1)
    assert torch.isfinite(full).all() and (norms - 1).abs().max().item() < 1e-5
    parts = torch.cat([enc.encode_ids(dids[s : s + 256].contiguous(), dlens[s : s + 256].contiguous()) for s in range(0, 1024, 256)])
    assert torch.equal(full, parts)
    again = enc.encode_ids(dids, dlens)
    assert torch.equal(full, again)
    # and a sample of it against the fp32 oracle
    sel = [0, 1, 2, 511, 1023]
    want = EO.forward(w, ids[sel], lens[sel], cfg).numpy()
    got = full[sel].cpu().numpy()
    cos = (want * got).sum(1) / (np.linalg.norm(want, axis=1) * np.linalg.norm(got, axis=1))
    assert cos.min() >= COS_TOL, cos


def test_encoder_matches_transformers_goldens(setup, golden_dir):
    """Against embeddings a transformers.BertModel produced from the same generated weights
    (tests/golden/make_encoder_goldens.py); tolerance: cosine >= 0.999."""
    eng, enc, w, cfg = setup
    g = np.load(golden_dir / "encoder_goldens.npz")
    assert int(g["seed"]) == 2026
    got = enc.encode_ids(torch.from_numpy(g["ids"]).cuda(), torch.from_numpy(g["lens"]).cuda()).cpu().numpy()
    want = g["embeddings"]
    cos = (want * got).sum(1) / (np.linalg.norm(want, axis=1) * np.linalg.norm(got, axis=1))
    assert cos.min() >= COS_TOL, cos


@pytest.mark.parametrize("gain", [7.0, 12.0])  # |q| |k| up to about 2.1 gain^2: around the limit, and far beyond it
def test_attention_with_large_scores_takes_the_row_maximum(setup, gain):
    """The attention kernel skips its row-maximum pass where |q| |k| bounds the scores below 2^100 (every case above:
    the generated weights score within +-2).  Query and key projections scaled up make the bound exceed that for some
    or all query tiles (gain^2 times the scores: softmax rows close to one-hot), so both forms run, side by side in one
    launch for the smaller gain; the result must stay with the oracle's."""
    eng, enc, w, cfg = setup
    w2 = dict(w)
    for l in range(cfg.layers):
        for n in ("q_w", "q_b", "k_w", "k_b"):
            w2[f"l{l}.{n}"] = np.asarray(w[f"l{l}.{n}"]) * gain
    enc2 = eng.encoder(w2, cfg)
    ids, lens = _inputs(12, 128, seed=int(gain))
    want = EO.forward(w2, ids, lens, cfg).numpy()
    got = enc2.encode_ids(torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    enc2.close()
    assert np.isfinite(got).all()
    cos = (got * want).sum(1)
    assert cos.min() >= COS_TOL, cos
