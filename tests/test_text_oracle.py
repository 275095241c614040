"""The text oracle against vectors produced by the reference itself (tests/golden/make_goldens.py)."""
import hashlib
import json

from oracle import text_oracle as T


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest()


def _sig(chunks):
    return [[c["chunk_index"], c["char_start"], c["char_end"], c["total_chunks"], sha(c["content"])] for c in chunks]


def test_chunk_corpus_63(text_goldens, corpus_texts):
    counts = []
    for name, want in text_goldens["chunk_corpus"].items():
        got = _sig(T.chunk(corpus_texts[name]))
        assert got == want, name
        counts.append(len(got))
    assert sorted(counts) == [6, 14, 15, 28] and sum(counts) == 63


def test_chunk_synthetic_shapes(text_goldens):
    for name, case in text_goldens["chunk_synth"].items():
        assert _sig(T.chunk(case["text"])) == case["chunks"], name


def test_chunk_params(text_goldens, corpus_texts):
    for case in text_goldens["chunk_params"]:
        got = [[c["char_start"], c["char_end"]] for c in T.chunk(corpus_texts["README.md"], case["size"], case["overlap"])]
        assert got == case["chunks"]


def test_clean(text_goldens):
    for src, want in text_goldens["clean"]:
        assert T.clean(src) == want


def test_doc_id(text_goldens, corpus_texts):
    for case in text_goldens["doc_id"]:
        content = case.get("content") or corpus_texts[case["name"]]
        assert T.doc_id(content, case["source"]) == case["id"]


def test_metadata(text_goldens, corpus_texts):
    for case in text_goldens["metadata"]:
        content = case["content"] if case["content"] is not None else corpus_texts[case["corpus_name"]]
        assert sha(content) == case["content_sha"]
        assert T.content_metadata(content, case["filepath"]) == case["md"]


def test_keywords(text_goldens):
    for q, want in text_goldens["keywords"]:
        assert T.keywords(q) == want, q


def test_keyword_score(text_goldens, corpus_texts):
    docs = [c["content"] for n in ["README.md", "automation-guide.md", "bash-linux-reference.md", "windows-powershell-reference.md"] for c in T.chunk(corpus_texts[n])]
    assert [sha(d) for d in docs] == text_goldens["docs_sha"]
    by_sha = {sha(d): d for d in docs}
    for case in text_goldens["keyword_score"]:
        doc = case["doc"] if case["doc"] is not None else by_sha[case["doc_sha"]]
        assert T.keyword_score(doc, case["kws"]) == case["score"]


def _docs(corpus_texts):
    return [c["content"] for n in ["README.md", "automation-guide.md", "bash-linux-reference.md", "windows-powershell-reference.md"] for c in T.chunk(corpus_texts[n])]


def test_hybrid_fusion_bit_exact(text_goldens, corpus_texts):
    docs = _docs(corpus_texts)
    for case in text_goldens["hybrid"]:
        rows, n = case["rows"], case["n"]
        k = min(2 * n, len(rows))
        assert case["stub_calls"][0]["n_results"] == 2 * n
        res = T.fuse(
            [docs[r] for r in rows[:k]],
            [{"source": f"doc{r}.md", "row": r} for r in rows[:k]],
            [f"id-{r:04d}" for r in rows[:k]],
            case["distances"][:k],
            case["query"],
            n,
        )
        assert len(res) == len(case["result"])
        for got, want in zip(res, case["result"]):
            assert got["id"] == want["id"]
            assert got["semantic_score"] == want["semantic_score"]
            assert isinstance(got["semantic_score"], int) == want["semantic_is_int"]
            assert got["keyword_score"] == want["keyword_score"]
            assert got["combined_score"] == want["combined_score"]
            assert got["distance"] == want["distance"]
            assert sha(got["document"]) == want["doc_sha"]
        assert json.dumps(T.bridge_hybrid(res, 800)) == case["bridge_json"]
        assert json.dumps(T.bridge_hybrid(res, 100)) == case["bridge_json_100"]
        assert json.dumps(res, indent=2) == case["cli_json"]


def test_plain_bridge_format(text_goldens, corpus_texts):
    docs = _docs(corpus_texts)
    for case in text_goldens["plain"]:
        rows, n = case["rows"], case["n"]
        k = min(n, len(rows))
        res = {
            "ids": [[f"id-{r:04d}" for r in rows[:k]]],
            "documents": [[docs[r] for r in rows[:k]]],
            "metadatas": [[{"source": f"doc{r}.md", "row": r} for r in rows[:k]]],
            "distances": [case["distances"][:k]],
        }
        assert json.dumps(T.bridge_plain(res, 800)) == case["bridge_json"]
        assert json.dumps(res, indent=2) == case["cli_json"]


def test_error_path(text_goldens):
    assert json.dumps(T.bridge_plain(text_goldens["error"]["result"])) == text_goldens["error"]["bridge_json"]
