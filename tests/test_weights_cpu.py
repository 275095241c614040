"""The local-checkpoint loader (the stand-in for SentenceTransformerEmbeddingFunction(model_name=...) fetching
all-MiniLM-L6-v2, scripts/chromadb_helper.py:64-66): weights written under BERT's own key names with
safetensors come back tensor for tensor, whatever prefix the exporting library used."""
import numpy as np
import pytest

from hybrid_rag_system_amd import weights as W

safetensors_numpy = pytest.importorskip("safetensors.numpy")

CFG = W.EncoderConfig(vocab=211, layers=2, max_pos=64)  # the same tensor list, small enough for a unit test


@pytest.mark.parametrize("prefix", ["", "bert.", "0.auto_model."])
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_load_safetensors_round_trip(tmp_path, prefix, dtype):
    gen = W.generate(CFG, 77)
    blob = {prefix + W.hf_name(name): t.astype(dtype) for name, t in gen.items()}
    blob[prefix + "pooler.dense.weight"] = np.zeros((4, 4), dtype)  # tensors the encoder does not use are ignored
    blob[prefix + "embeddings.position_ids"] = np.arange(CFG.max_pos, dtype=np.int64)[None]
    safetensors_numpy.save_file(blob, str(tmp_path / "model.safetensors"))
    for path in (str(tmp_path), str(tmp_path / "model.safetensors")):  # a directory or the file itself
        got = W.load_safetensors(path, CFG)
        assert list(got) == [n for n, _ in W.tensor_specs(CFG)]
        for name, shape in W.tensor_specs(CFG):
            assert got[name].dtype == np.float32 and got[name].shape == tuple(shape) and got[name].flags["C_CONTIGUOUS"]
            assert np.array_equal(got[name], gen[name].astype(dtype).astype(np.float32)), name


def test_load_safetensors_reports_what_is_wrong(tmp_path):
    gen = W.generate(CFG, 78)
    blob = {W.hf_name(name): t for name, t in gen.items()}
    missing = dict(blob)
    del missing["encoder.layer.1.output.dense.weight"]
    safetensors_numpy.save_file(missing, str(tmp_path / "a.safetensors"))
    with pytest.raises(KeyError, match="encoder.layer.1.output.dense.weight"):
        W.load_safetensors(str(tmp_path / "a.safetensors"), CFG)
    bad = dict(blob)
    bad["encoder.layer.0.intermediate.dense.weight"] = np.zeros((CFG.hidden, CFG.inter), np.float32)  # transposed
    safetensors_numpy.save_file(bad, str(tmp_path / "b.safetensors"))
    with pytest.raises(ValueError, match="intermediate.dense.weight"):
        W.load_safetensors(str(tmp_path / "b.safetensors"), CFG)


def test_hf_names_are_the_ones_bertmodel_uses():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.BertConfig(vocab_size=CFG.vocab, hidden_size=CFG.hidden, num_hidden_layers=CFG.layers, num_attention_heads=CFG.heads,
                                  intermediate_size=CFG.inter, max_position_embeddings=CFG.max_pos)
    keys = set(transformers.BertModel(cfg, add_pooling_layer=False).state_dict())
    ours = {W.hf_name(n) for n, _ in W.tensor_specs(CFG)}
    assert ours <= keys and all("position_ids" in k or k in ours for k in keys)
