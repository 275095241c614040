"""The encoder oracle against a config-constructed transformers.BertModel with the same weights."""
import numpy as np
import pytest
import torch

from hybrid_rag_system_amd import weights as W
from oracle import encoder_oracle as EO


def _inputs(B=4, S=64, seed=0):
    rng = np.random.default_rng(seed)
    ids = rng.integers(1000, 30000, (B, S))
    lens = rng.integers(S // 4, S + 1, B)
    lens[0] = S
    ids[:, 0] = 101
    for b in range(B):
        ids[b, lens[b] - 1] = 102
        ids[b, lens[b]:] = 0
    return ids, lens


def test_generator_is_deterministic_and_shaped():
    cfg = W.EncoderConfig()
    w = W.generate(cfg, seed=7)
    specs = W.tensor_specs(cfg)
    assert len(specs) == 101 and sum(int(np.prod(s)) for _, s in specs) == 22_565_376
    assert all(w[n].shape == s and w[n].dtype == np.float32 for n, s in specs)
    w2 = W.generate(cfg, seed=7)
    assert all(np.array_equal(w[n], w2[n]) for n, _ in specs)
    assert abs(float(w["l0.q_w"].std()) - 0.6 / np.sqrt(384) / np.sqrt(3)) < 1e-3
    assert float(w["word_emb"][:4].sum()) == float(W.generate(cfg, seed=7)["word_emb"][:4].sum())


def test_oracle_matches_transformers_bert():
    transformers = pytest.importorskip("transformers")
    cfg = W.EncoderConfig()
    w = W.generate(cfg, seed=2026)
    hf_cfg = transformers.BertConfig(
        vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
        intermediate_size=cfg.inter, max_position_embeddings=cfg.max_pos, type_vocab_size=2, layer_norm_eps=cfg.eps, hidden_act="gelu",
    )
    model = transformers.BertModel(hf_cfg, add_pooling_layer=False).eval()
    sd = model.state_dict()
    for name, _ in W.tensor_specs(cfg):
        key = W.hf_name(name)
        assert key in sd, key
        sd[key] = torch.from_numpy(w[name])
    model.load_state_dict(sd, strict=False)
    ids, lens = _inputs()
    mask = (np.arange(ids.shape[1])[None] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state
        m = torch.from_numpy(mask).float()
        pooled = (out * m[..., None]).sum(1) / m.sum(1, keepdim=True).clamp(min=1e-9)
        want = torch.nn.functional.normalize(pooled, p=2, dim=1)
        got, hidden = EO.forward(w, ids, lens, cfg, return_hidden=True)
    assert torch.allclose(hidden[-1] * m[..., None], out * m[..., None], atol=2e-5)
    assert torch.allclose(got, want, atol=1e-6)
    assert float((got * want).sum(1).min()) > 0.999999
    # the embeddings of different inputs differ (the random network is not degenerate)
    assert float((got[0] * got[1]).sum()) < 0.999


def test_oracle_matches_committed_transformers_goldens(golden_dir):
    """tests/golden/encoder_goldens.npz was produced by transformers.BertModel (make_encoder_goldens.py)."""
    g = np.load(golden_dir / "encoder_goldens.npz")
    cfg = W.EncoderConfig()
    w = W.generate(cfg, seed=int(g["seed"]))
    got, hidden = EO.forward(w, g["ids"], g["lens"], cfg, return_hidden=True)
    assert np.abs(got.numpy() - g["embeddings"]).max() < 2e-6
    assert len(hidden) == cfg.layers + 1  # embedding output, then every layer
    assert np.abs(hidden[1][:, :4, :16].numpy() - g["hidden_layer0"]).max() < 2e-5
    assert np.abs(hidden[-1][:, :4, :16].numpy() - g["hidden_final"]).max() < 5e-5
