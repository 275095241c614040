#!/usr/bin/env python3
"""Golden vectors for the encoder: a transformers.BertModel built from a CONFIG OBJECT (MiniLM-L6
geometry, no from_pretrained, nothing downloaded) filled with this repo's generated weights, run on
fixed ragged inputs; masked mean pool + L2 normalise as sentence-transformers does behind
scripts/chromadb_helper.py:64-66.  Writes encoder_goldens.npz (ids, lens, embeddings, a slice of the
layer-0 and final hidden states).  Run from the repo root: python tests/golden/make_encoder_goldens.py"""
import os
import sys

import numpy as np
import torch
import transformers

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hybrid_rag_system_amd import weights as W  # noqa: E402

SEED = 2026
cfg = W.EncoderConfig()
w = W.generate(cfg, seed=SEED)
hf_cfg = transformers.BertConfig(
    vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
    intermediate_size=cfg.inter, max_position_embeddings=cfg.max_pos, type_vocab_size=2, layer_norm_eps=cfg.eps, hidden_act="gelu",
)
model = transformers.BertModel(hf_cfg, add_pooling_layer=False).eval()
sd = model.state_dict()
for name, _ in W.tensor_specs(cfg):
    sd[W.hf_name(name)] = torch.from_numpy(w[name])
model.load_state_dict(sd, strict=False)

rng = np.random.default_rng(808)
B, S = 8, 256
ids = rng.integers(1000, 30000, (B, S)).astype(np.int32)
lens = np.array([256, 1 + 1, 37, 100, 255, 64, 200, 129], np.int32)
ids[:, 0] = 101
for b in range(B):
    ids[b, lens[b] - 1] = 102
    ids[b, lens[b]:] = 0
mask = (np.arange(S)[None] < lens[:, None]).astype(np.int64)
with torch.no_grad():
    out = model(input_ids=torch.from_numpy(ids.astype(np.int64)), attention_mask=torch.from_numpy(mask), output_hidden_states=True)
    m = torch.from_numpy(mask).float()
    pooled = (out.last_hidden_state * m[..., None]).sum(1) / m.sum(1, keepdim=True).clamp(min=1e-9)
    emb = torch.nn.functional.normalize(pooled, p=2, dim=1)
np.savez_compressed(
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "encoder_goldens.npz"),
    seed=np.int64(SEED), ids=ids, lens=lens, embeddings=emb.numpy().astype(np.float32),
    hidden_layer0=out.hidden_states[1][:, :4, :16].numpy().astype(np.float32),
    hidden_final=out.last_hidden_state[:, :4, :16].numpy().astype(np.float32),
    transformers_version=np.array(transformers.__version__),
)
print("wrote encoder_goldens.npz", emb.shape)
