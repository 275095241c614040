"""Encoder oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py).

fp32 restatement of what `SentenceTransformerEmbeddingFunction(model_name="all-MiniLM-L6-v2")`
(call site scripts/chromadb_helper.py:58-71) computes after tokenisation: the BERT/MiniLM-L6 forward
(post-LayerNorm encoder, erf-GELU, additive attention mask), masked mean pooling and L2 normalisation.
sentence-transformers and the model weights are absent and cannot be fetched: PARITY UNPINNED against
the real model; this file is cross-checked in the build container against a config-constructed
`transformers.BertModel` holding the same (generated) weights (tests/test_encoder_oracle.py).
"""
import math

import torch
import torch.nn.functional as F


def forward(w, ids, lens, cfg, return_hidden=False):
    """w: dict name -> fp32 tensor/ndarray (hybrid_rag_system_amd.weights order); ids int [B,S]; lens int [B]."""
    t = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in w.items()}
    ids = torch.as_tensor(ids, dtype=torch.long)
    lens = torch.as_tensor(lens, dtype=torch.long)
    B, S = ids.shape
    H, nh = cfg.hidden, cfg.heads
    dh = H // nh
    mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.float32)  # [B,S]
    x = t["word_emb"][ids] + t["pos_emb"][:S][None] + t["type_emb"][0][None, None]
    x = F.layer_norm(x, (H,), t["emb_ln_g"], t["emb_ln_b"], cfg.eps)
    bias = (1.0 - mask)[:, None, None, :] * torch.finfo(torch.float32).min
    hidden = [x]
    for l in range(cfg.layers):
        p = f"l{l}."
        q = (x @ t[p + "q_w"].T + t[p + "q_b"]).view(B, S, nh, dh).transpose(1, 2)
        k = (x @ t[p + "k_w"].T + t[p + "k_b"]).view(B, S, nh, dh).transpose(1, 2)
        v = (x @ t[p + "v_w"].T + t[p + "v_b"]).view(B, S, nh, dh).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(dh) + bias
        a = torch.softmax(s, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, S, H)
        x = F.layer_norm(x + a @ t[p + "o_w"].T + t[p + "o_b"], (H,), t[p + "ln1_g"], t[p + "ln1_b"], cfg.eps)
        h = F.gelu(x @ t[p + "i_w"].T + t[p + "i_b"])
        x = F.layer_norm(x + h @ t[p + "d_w"].T + t[p + "d_b"], (H,), t[p + "ln2_g"], t[p + "ln2_b"], cfg.eps)
        hidden.append(x)
    pooled = (x * mask[..., None]).sum(1) / mask.sum(1, keepdim=True).clamp(min=1e-9)
    emb = F.normalize(pooled, p=2, dim=1)
    return (emb, hidden) if return_hidden else emb
