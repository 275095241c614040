"""MiniLM-L6 geometry and weight sources for the encoder.

The reference loads `all-MiniLM-L6-v2` by NAME through sentence-transformers
(scripts/chromadb_helper.py:58-71), i.e. from the Hugging Face hub.  No weights or vocabulary exist
offline, so two sources are offered: a local directory with `model.safetensors` (real weights,
BERT key names), or a deterministic counter-based generator (same numbers on every machine; used by
tests and benchmarks, where only the arithmetic matters).

Tensor order (101 tensors, the order of `hrs_weights` in include/hrs.h):
  word_emb [V,H], pos_emb [P,H], type_emb [2,H], emb_ln_g [H], emb_ln_b [H], then per layer
  q_w [H,H], q_b, k_w, k_b, v_w, v_b, o_w [H,H], o_b, ln1_g, ln1_b, i_w [I,H], i_b [I], d_w [H,I], d_b, ln2_g, ln2_b
Linear weights are [out, in] (y = x @ w.T + b), as BERT stores them.
"""
import os
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class EncoderConfig:
    vocab: int = 30522
    hidden: int = 384
    layers: int = 6
    heads: int = 12
    inter: int = 1536
    max_pos: int = 512
    eps: float = 1e-12
    max_seq: int = 256  # sentence-transformers truncates all-MiniLM-L6-v2 inputs to 256 tokens


def tensor_specs(cfg):
    H, I = cfg.hidden, cfg.inter
    specs = [("word_emb", (cfg.vocab, H)), ("pos_emb", (cfg.max_pos, H)), ("type_emb", (2, H)), ("emb_ln_g", (H,)), ("emb_ln_b", (H,))]
    for l in range(cfg.layers):
        p = f"l{l}."
        specs += [
            (p + "q_w", (H, H)), (p + "q_b", (H,)), (p + "k_w", (H, H)), (p + "k_b", (H,)), (p + "v_w", (H, H)), (p + "v_b", (H,)),
            (p + "o_w", (H, H)), (p + "o_b", (H,)), (p + "ln1_g", (H,)), (p + "ln1_b", (H,)),
            (p + "i_w", (I, H)), (p + "i_b", (I,)), (p + "d_w", (H, I)), (p + "d_b", (H,)), (p + "ln2_g", (H,)), (p + "ln2_b", (H,)),
        ]
    return specs


def _mix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return x ^ (x >> np.uint64(31))


def _uniform(seed, tensor_index, n):
    """n uniforms in (-1, 1), keyed (seed, tensor, element) -- pure integer hashing, so bit-identical everywhere."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _mix(_mix(np.uint64(seed) ^ (np.uint64(tensor_index + 1) * np.uint64(0xD1B54A32D192ED03))) + idx)
    return ((h >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0).astype(np.float32)


def generate(cfg=EncoderConfig(), seed=2026):
    """Deterministic weights, uniform with per-kind scales chosen so that the random network behaves like an
    embedding model rather than collapsing every input onto one direction: word embeddings dominate the
    position / type embeddings and the biases (the part common to all inputs), and the sub-layers add to the
    residual stream without swamping it.  Unrelated random inputs then embed at cosine ~0.2 (with equal
    scales everywhere it is ~0.99 and bf16-vs-fp32 rank comparisons are meaningless)."""
    out = {}
    for t, (name, shape) in enumerate(tensor_specs(cfg)):
        n = int(np.prod(shape))
        u = _uniform(seed, t, n).reshape(shape)
        base = name.split(".")[-1]
        if base.endswith("_g"):
            w = 1.0 + 0.1 * u
        elif base.endswith("_b"):
            w = 0.01 * u
        elif base == "word_emb":
            w = 0.5 * u
        elif base in ("pos_emb", "type_emb"):
            w = 0.02 * u
        else:
            w = u * np.float32(0.6 / np.sqrt(shape[1]))  # var = 0.12 / fan_in
        out[name] = np.ascontiguousarray(w.astype(np.float32))
    return out


_HF_LAYER = {
    "q_w": "attention.self.query.weight", "q_b": "attention.self.query.bias",
    "k_w": "attention.self.key.weight", "k_b": "attention.self.key.bias",
    "v_w": "attention.self.value.weight", "v_b": "attention.self.value.bias",
    "o_w": "attention.output.dense.weight", "o_b": "attention.output.dense.bias",
    "ln1_g": "attention.output.LayerNorm.weight", "ln1_b": "attention.output.LayerNorm.bias",
    "i_w": "intermediate.dense.weight", "i_b": "intermediate.dense.bias",
    "d_w": "output.dense.weight", "d_b": "output.dense.bias",
    "ln2_g": "output.LayerNorm.weight", "ln2_b": "output.LayerNorm.bias",
}
_HF_TOP = {
    "word_emb": "embeddings.word_embeddings.weight", "pos_emb": "embeddings.position_embeddings.weight",
    "type_emb": "embeddings.token_type_embeddings.weight", "emb_ln_g": "embeddings.LayerNorm.weight", "emb_ln_b": "embeddings.LayerNorm.bias",
}


def hf_name(name):
    """Our tensor name -> the key BertModel.state_dict() uses."""
    if name in _HF_TOP:
        return _HF_TOP[name]
    layer, base = name.split(".")
    return f"encoder.layer.{int(layer[1:])}.{_HF_LAYER[base]}"


def load_safetensors(path, cfg=EncoderConfig()):
    """Real weights from a LOCAL model.safetensors (BERT key names, optional 'bert.' prefix)."""
    from safetensors.numpy import load_file

    if os.path.isdir(path):
        path = os.path.join(path, "model.safetensors")
    raw = load_file(path)
    out = {}
    for name, shape in tensor_specs(cfg):
        key = hf_name(name)
        for cand in (key, "bert." + key, "0.auto_model." + key):
            if cand in raw:
                w = raw[cand]
                break
        else:
            raise KeyError(f"{path}: tensor {key} not found")
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"{key}: shape {w.shape}, expected {shape}")
        out[name] = np.ascontiguousarray(w.astype(np.float32))
    return out
