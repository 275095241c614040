"""The real-checkpoint path end to end on the GPU: HRS_MODEL_DIR holds a `model.safetensors` (BERT key names,
MiniLM-L6 geometry) and a `vocab.txt`; `index` and `query --hybrid` then run on the loaded weights with the native
WordPiece tokenizer -- what SentenceTransformerEmbeddingFunction(model_name="all-MiniLM-L6-v2") stands for in the
reference (scripts/chromadb_helper.py:64-66).  The real checkpoint cannot be fetched offline, so the directory is
written by the test (generated weights under another seed, a vocabulary made from the fixture corpus): parity with
the real model stays unpinned, the loader / tokenizer / encoder chain is what is exercised."""
import importlib
import os
import re
import zlib

import numpy as np
import pytest

from oracle import encoder_oracle as EO
from oracle import search_oracle as so
from oracle import text_oracle as T

pytestmark = pytest.mark.gpu

CORPUS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "context", "infrastructure")
QUERY = "How do I restart the nginx container on Unraid?"


def _write_model_dir(path):
    from safetensors.numpy import save_file

    from hybrid_rag_system_amd import weights as W

    cfg = W.EncoderConfig()
    gen = W.generate(cfg, 31337)
    save_file({"bert." + W.hf_name(n): t for n, t in gen.items()}, os.path.join(path, "model.safetensors"))
    # a bert-base-uncased shaped vocabulary: specials at their usual ids, then whole words and ##pieces of the corpus
    words = set()
    for name in sorted(os.listdir(CORPUS_DIR)):
        with open(os.path.join(CORPUS_DIR, name), encoding="utf-8", errors="ignore") as f:
            words.update(re.findall(r"[a-z]+", f.read().lower()))
    words = sorted(words)
    pieces = [w for w in words if len(w) <= 6][:4000]
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab += list("abcdefghijklmnopqrstuvwxyz0123456789") + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    vocab += list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")
    vocab += pieces + ["##" + p for p in pieces if len(p) <= 4] + [w for w in words if len(w) > 6 and zlib.crc32(w.encode()) % 3]  # a third of the long words must split
    seen, out = set(), []
    for v in vocab:
        if v not in seen:
            seen.add(v)
            out.append(v)
    out += [f"[unused{i}]" for i in range(1000, 1000 + cfg.vocab - len(out))]
    assert len(out) == cfg.vocab
    with open(os.path.join(path, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(out) + "\n")
    return cfg, gen


def test_index_and_hybrid_query_from_a_local_checkpoint(tmp_path, monkeypatch):
    model_dir = tmp_path / "model"
    model_dir.mkdir()
    cfg, gen = _write_model_dir(str(model_dir))
    monkeypatch.setenv("HRS_MODEL_DIR", str(model_dir))
    monkeypatch.setenv("HRS_DATA_DIR", str(tmp_path / "data"))
    from hybrid_rag_system_amd import helper as h
    from hybrid_rag_system_amd.tokenizer import WordPieceTokenizer

    importlib.reload(h)
    try:
        ef = h.get_embedding_function()
        assert ef.source == str(model_dir) and isinstance(ef.tokenizer, WordPieceTokenizer)
        res = h.index_directory(CORPUS_DIR, ["*.md"], "from-checkpoint")
        assert res["status"] == "success" and res["indexed"] == 4
        col = h.get_collection("from-checkpoint")
        assert col.count() == 63
        # the native WordPiece tokenizer gives the ids of the Python definition, with real sub-word splits
        ids, lens = ef.tokenizer.encode_batch(col.documents)
        pid, plen = ef.tokenizer.encode_batch_python(col.documents)
        assert np.array_equal(ids, pid) and np.array_equal(lens, plen)
        unk = ef.tokenizer.unk_id
        assert (ids == unk).mean() < 0.05 and (ids >= ef.tokenizer.vocab["##a"]).any()
        # embeddings: the loaded weights through the HIP encoder against the fp32 oracle on the same ids
        want = EO.forward(gen, ids, lens, cfg).numpy()
        cos = (want * col._emb).sum(1)
        assert cos.min() >= 0.999, cos.min()
        other = EO.forward(__import__("hybrid_rag_system_amd.weights", fromlist=["generate"]).generate(cfg, 2026), ids, lens, cfg).numpy()
        assert (other * col._emb).sum(1).max() < 0.9  # ... and they are not the default generated weights'
        # query --hybrid: search on identical fp32 embeddings is bit-exact, fusion is the reference's arithmetic
        got = h.hybrid_query(QUERY, 5, "from-checkpoint")
        qe = h.embed([QUERY]).cpu().numpy()
        wd, wr = so.topk(qe, col._emb, 10, "l2")
        want_rows = T.fuse([col.documents[r] for r in wr[0]], [col.metadatas[r] for r in wr[0]], [col.ids[r] for r in wr[0]], [float(x) for x in wd[0]], QUERY, 5)
        assert got == want_rows and len(got) == 5
    finally:
        monkeypatch.delenv("HRS_MODEL_DIR")
        monkeypatch.delenv("HRS_DATA_DIR")
        importlib.reload(h)
