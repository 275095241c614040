"""The helper / bridge mirrors against vectors produced by the reference itself (stub collection
in place of the vector store, exactly as tests/golden/make_goldens.py ran the reference)."""
import contextlib
import hashlib
import io
import json

import pytest

from hybrid_rag_system_amd import bridge, helper, textproc
from hybrid_rag_system_amd.tokenizer import HashTokenizer, WordPieceTokenizer, basic_tokenize

NAMES = ["README.md", "automation-guide.md", "bash-linux-reference.md", "windows-powershell-reference.md"]


def sha(s):
    return hashlib.sha256(s.encode()).hexdigest()


class Stub:
    name, metadata = "stub", {"description": "stub"}

    def __init__(self, docs, metas, ids, dists):
        self.rows, self.calls = (docs, metas, ids, dists), []

    def query(self, query_texts, n_results, include, where=None):
        self.calls.append(dict(n_results=n_results, include=list(include), where=where))
        docs, metas, ids, dists = self.rows
        k = min(n_results, len(docs))
        return {"ids": [ids[:k]], "documents": [docs[:k]], "metadatas": [metas[:k]], "distances": [dists[:k]]}


@pytest.fixture()
def docs(corpus_texts):
    return [c["content"] for n in NAMES for c in helper.chunk_text(corpus_texts[n])]


def test_text_functions_match_reference(text_goldens, corpus_texts, docs):
    for name, want in text_goldens["chunk_corpus"].items():
        got = [[c["chunk_index"], c["char_start"], c["char_end"], c["total_chunks"], sha(c["content"])] for c in helper.chunk_text(corpus_texts[name])]
        assert got == want
    assert len(docs) == 63 and [sha(d) for d in docs] == text_goldens["docs_sha"]
    for name, case in text_goldens["chunk_synth"].items():
        got = [[c["chunk_index"], c["char_start"], c["char_end"], c["total_chunks"], sha(c["content"])] for c in helper.chunk_text(case["text"])]
        assert got == case["chunks"], name
    for case in text_goldens["chunk_params"]:
        assert [[c["char_start"], c["char_end"]] for c in helper.chunk_text(corpus_texts["README.md"], case["size"], case["overlap"])] == case["chunks"]
    for src, want in text_goldens["clean"]:
        assert helper.clean_text(src) == want
    for case in text_goldens["doc_id"]:
        assert helper.generate_doc_id(case.get("content") or corpus_texts[case["name"]], case["source"]) == case["id"]
    for case in text_goldens["metadata"]:
        content = case["content"] if case["content"] is not None else corpus_texts[case["corpus_name"]]
        assert helper.extract_metadata_from_content(content, case["filepath"]) == case["md"]
    for q, want in text_goldens["keywords"]:
        assert sorted(helper.extract_keywords(q)) == want
    by_sha = {sha(d): d for d in docs}
    for case in text_goldens["keyword_score"]:
        doc = case["doc"] if case["doc"] is not None else by_sha[case["doc_sha"]]
        assert helper.calculate_keyword_score(doc, case["kws"]) == case["score"]
    assert len(textproc.STOPWORDS) == 101


def test_hybrid_query_bit_exact_and_json_contracts(text_goldens, docs, monkeypatch):
    for case in text_goldens["hybrid"]:
        rows = case["rows"]
        stub = Stub([docs[r] for r in rows], [{"source": f"doc{r}.md", "row": r} for r in rows], [f"id-{r:04d}" for r in rows], case["distances"])
        monkeypatch.setattr(helper, "get_collection", lambda name=None, _s=stub: _s)
        res = helper.hybrid_query(case["query"], case["n"], "whatever")
        assert stub.calls == case["stub_calls"]
        assert len(res) == len(case["result"])
        for got, want in zip(res, case["result"]):
            assert (got["id"], got["metadata"], got["distance"]) == (want["id"], want["metadata"], want["distance"])
            assert got["semantic_score"] == want["semantic_score"] and isinstance(got["semantic_score"], int) == want["semantic_is_int"]
            assert got["keyword_score"] == want["keyword_score"] and got["combined_score"] == want["combined_score"]
            assert sha(got["document"]) == want["doc_sha"]
        assert json.dumps(bridge.format_hybrid_results(res, 800)) == case["bridge_json"]
        assert json.dumps(bridge.format_hybrid_results(res, 100)) == case["bridge_json_100"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            helper.main(["query", case["query"], "--hybrid", "--json", "-n", str(case["n"])])
        assert buf.getvalue() == case["cli_json"] + "\n"
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bridge.main(["query", case["query"], "--hybrid", "--n-results", str(case["n"])])
        assert buf.getvalue() == case["bridge_json"] + "\n"


def test_plain_query_and_error_path(text_goldens, docs, monkeypatch):
    for case in text_goldens["plain"]:
        rows = case["rows"]
        stub = Stub([docs[r] for r in rows], [{"source": f"doc{r}.md", "row": r} for r in rows], [f"id-{r:04d}" for r in rows], case["distances"])
        monkeypatch.setattr(helper, "get_collection", lambda name=None, _s=stub: _s)
        res = helper.query(case["query"], case["n"], "c", {"system": "docker"} if case["n"] == 3 else None)
        assert stub.calls == case["stub_calls"]
        assert json.dumps(bridge.format_query_results(res, 800)) == case["bridge_json"]
        assert json.dumps(res, indent=2) == case["cli_json"]

    class Boom:
        def query(self, **kw):
            raise RuntimeError("collection exploded")

    monkeypatch.setattr(helper, "get_collection", lambda name=None: Boom())
    err = helper.query("x")
    assert err == text_goldens["error"]["result"]
    assert json.dumps(bridge.format_query_results(err)) == text_goldens["error"]["bridge_json"]
    with pytest.raises(RuntimeError):  # hybrid_query does not catch, as in the reference
        helper.hybrid_query("x")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), pytest.raises(SystemExit) as ex:
        bridge.main(["query", "x", "--hybrid"])
    assert ex.value.code == 1 and json.loads(buf.getvalue()) == {"success": False, "error": "collection exploded"}


def test_cli_text_output(text_goldens, docs, monkeypatch):
    case = text_goldens["cli_text_hybrid"]
    rows = case["rows"]
    stub = Stub([docs[r] for r in rows], [{"source": f"doc{r}.md", "system": "docker"} if r % 2 else {} for r in rows], [f"id-{r}" for r in rows], case["distances"])
    monkeypatch.setattr(helper, "get_collection", lambda name=None, _s=stub: _s)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        helper.main(["query", case["query"], "--hybrid", "-n", str(case["n"])])
    assert buf.getvalue() == case["stdout"]
    # plain text output works (the reference raises on this path because of a malformed format spec)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        helper.main(["query", case["query"], "-n", "2"])
    assert "--- Result 1 (distance: 0.310) ---" in buf.getvalue()


def test_index_and_add_status_dicts(tmp_path, monkeypatch):
    added = []

    class Col:
        name, metadata = "c", None

        def add(self, documents, metadatas, ids):
            added.append((documents, metadatas, ids))

        def count(self):
            return sum(len(a[2]) for a in added)

        def peek(self, limit=10):
            return {"metadatas": [m for a in added for m in a[1]][:limit]}

    monkeypatch.setattr(helper, "get_collection", lambda name=None: Col())
    assert helper.index_file(str(tmp_path / "missing.md")) == {"status": "error", "error": f"File not found: {tmp_path / 'missing.md'}"}
    big = tmp_path / "guide.md"
    big.write_text("# Title\n" + "docker compose word. " * 200)
    r = helper.index_file(str(big))
    assert r["status"] == "success" and r["chunks"] == len(added[-1][2]) > 1
    docs_, metas, ids = added[-1]
    assert ids[0].endswith("-0000") and len(ids[0]) == 21 and metas[0]["source"] == "guide.md" and metas[0]["category"] == "howto"
    assert metas[0]["system"] == "docker" and metas[1]["chunk_index"] == 1 and metas[0]["title"] == "Title"
    small = tmp_path / "note.txt"
    small.write_text("short unraid note")
    assert helper.index_file(str(small))["chunks"] == 1 and "chunk_index" not in added[-1][1][0]
    out = helper.index_directory(str(tmp_path))
    assert out["status"] == "success" and out["indexed"] == 2 and out["failed"] == 0
    assert helper.index_directory(str(tmp_path / "nope"))["status"] == "error"
    r = helper.add_document("some unraid text", {"source": "s.md"})
    assert r == {"status": "success", "id": helper.generate_doc_id("some unraid text", "s.md")}
    assert added[-1][1][0]["system"] == "unraid" and "indexed_at" in added[-1][1][0]
    st = helper.get_collection_stats()
    assert st["name"] == "c" and st["count"] > 0 and "docker" in st["systems_sample"]


def test_wordpiece_matches_transformers(corpus_texts, tmp_path):
    transformers = pytest.importorskip("transformers")
    words = sorted({w for t in corpus_texts.values() for w in basic_tokenize(t)})
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    # whole words for half of them, 3-char pieces for the rest, single characters as a floor
    vocab += words[::2]
    vocab += sorted({w[:3] for w in words} | {"##" + w[i : i + 3] for w in words for i in range(3, len(w), 3)})
    vocab += sorted({c for w in words for c in w} | {"##" + c for w in words for c in w})
    vocab = list(dict.fromkeys(vocab))
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(vocab) + "\n")
    ref = transformers.BertTokenizer(str(vf), do_lower_case=True)
    mine = WordPieceTokenizer(str(vf))
    samples = [c["content"] for n in NAMES for c in helper.chunk_text(corpus_texts[n])][::4] + ["Héllo wörld — naïve café!", "", "x" * 150, "漢字 mixed 文字"]
    for text in samples:
        want = ref(text, truncation=True, max_length=256)["input_ids"]
        assert mine.encode(text) == want
    ids, lens = mine.encode_batch(samples[:5])
    assert ids.shape[1] % 32 == 0 and ids.shape[1] <= 256 and (lens <= ids.shape[1]).all()


def test_hash_tokenizer_is_stable():
    t = HashTokenizer()
    a = t.encode("Restart the nginx container, please!")
    assert a[0] == 101 and a[-1] == 102 and len(a) == 9 and all(1000 <= x < 30522 for x in a[1:-1])
    assert a == t.encode("restart  THE nginx container , please !")
    assert len(t.encode("word " * 1000)) == 256


def test_clean_text_equals_the_regex_form():
    """clean_text uses str.split(); the reference uses re.sub(r"\\s+", " ", ...) (chromadb_helper.py:93-99)."""
    import re

    from hybrid_rag_system_amd.textproc import clean_text

    ws = re.compile(r"\s+")
    for cp in list(range(0x3100)) + [0xFEFF, 0x1D7FF, 0xE0020]:
        ch = chr(cp)
        for t in (f"a{ch}b", f"{ch}a {ch}{ch} b{ch}", f"\x00{ch} a", ch):
            assert clean_text(t) == ws.sub(" ", t).replace("\x00", "").strip(), hex(cp)


def test_where_forms_of_the_reference_callers():
    """scripts/chromadb_helper.py:265-266, scripts/hybrid-rag.py:358-370, scripts/improved-router.py:463-477."""
    from hybrid_rag_system_amd.store import _match

    m = {"system": "unraid", "topics": "docker,networking,backup", "chunk_index": 3}
    assert _match(m, {"system": "unraid"}) and not _match(m, {"system": "proxmox"}) and not _match(None, {"system": "unraid"})
    assert _match(m, {"$and": [{"system": "unraid"}, {"topics": {"$contains": "networking"}}]})
    assert not _match(m, {"$and": [{"system": "unraid"}, {"topics": {"$contains": "kubernetes"}}]})
    assert _match(m, {"topics": {"$contains": "backup"}}) and not _match({"system": "x"}, {"topics": {"$contains": "backup"}})
    assert _match(m, {"$or": [{"system": "proxmox"}, {"system": "unraid"}]}) and not _match(m, {"$or": [{"system": "proxmox"}, {"system": "windows"}]})
    assert _match(m, {"system": {"$in": ["unraid", "docker"]}}) and _match(m, {"system": {"$nin": ["docker"]}}) and _match(m, {"system": {"$ne": "docker"}})
    assert _match(m, {"chunk_index": {"$gte": 3, "$lt": 4}}) and not _match(m, {"chunk_index": {"$gt": 3}}) and not _match(m, {"system": {"$gt": 1}})
    assert _match({}, {"system": {"$ne": "x"}}) and not _match({}, {"system": {"$eq": "x"}})
    with pytest.raises(ValueError, match="unsupported where operator"):
        _match(m, {"system": {"$regex": "u.*"}})


def test_chunk_document_matches_the_reference():
    """The multi-system indexer's chunker against chunks produced by the reference's own function
    (tests/golden/make_chunk_document_goldens.py): count, lengths and hashes of every chunk."""
    import hashlib
    import json
    import os

    from hybrid_rag_system_amd.textproc import chunk_document
    from tests.chunk_document_cases import build_cases

    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chunk_document_goldens.json")))
    cases = build_cases()
    assert set(cases) == set(want) and len(cases) >= 30
    for name, (text, kw) in cases.items():
        got = chunk_document(text, **kw)
        assert len(got) == want[name]["n"], name
        assert [len(c) for c in got] == want[name]["lens"], name
        assert [hashlib.sha256(c.encode()).hexdigest()[:16] for c in got] == want[name]["sha"], name


def test_bm25_keyword_score_properties():
    """The BM25 keyword weighting (an extension; the reference's score is the fraction): in [0, 1), zero without a
    match, grows with the term frequency, falls with the document's length, and a rare keyword outweighs a common one."""
    from hybrid_rag_system_amd.textproc import bm25_keyword_score, calculate_keyword_score

    df, n, avg = {"proxmox": 3, "server": 400, "backup": 50}, 1000, 100.0
    kws = ["proxmox", "server", "backup"]
    s = lambda doc: bm25_keyword_score(doc, kws, df, n, avg)
    assert s("nothing relevant here") == 0.0 and bm25_keyword_score("proxmox", [], df, n, avg) == 0.0
    one, two = s("the proxmox host"), s("the proxmox host runs proxmox")
    assert 0.0 < one < two < 1.0
    assert s("proxmox " + "filler " * 400) < one
    assert s("a proxmox") > s("a server")  # rarity
    assert calculate_keyword_score("a Proxmox SERVER", kws) == 2 / 3  # the reference's fraction is untouched
    assert 0.0 < s("Proxmox SERVER backup " * 50) < 1.0
