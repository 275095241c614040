"""BASELINE config 1 end to end on the GPU: index context/infrastructure/*.md (63 chunks), then
`query --hybrid --n-results 5`, through the helper / bridge surface."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from oracle import encoder_oracle as EO
from oracle import search_oracle as so
from oracle import text_oracle as T

pytestmark = pytest.mark.gpu

CORPUS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "context", "infrastructure")
QUERIES = [
    "How do I restart the nginx container on Unraid?",
    "powershell Get-ADUser filter examples",
    "bash loop over files in a directory",
    "proxmox lxc container backup schedule",
    "the and of to in",
]


@pytest.fixture(scope="module")
def helper(tmp_path_factory):
    data = tmp_path_factory.mktemp("hrs_data")
    os.environ["HRS_DATA_DIR"] = str(data)
    import importlib

    from hybrid_rag_system_amd import helper as h

    importlib.reload(h)
    yield h


def test_index_then_hybrid_query(helper):
    res = helper.index_directory(CORPUS_DIR, ["*.md"], "cfg1")
    assert res["status"] == "success" and res["indexed"] == 4 and res["failed"] == 0
    assert sorted(f["chunks"] for f in res["files"]) == [6, 14, 15, 28]
    col = helper.get_collection("cfg1")
    assert col.count() == 63
    # indexing again adds nothing: existing ids are kept
    helper.index_directory(CORPUS_DIR, ["*.md"], "cfg1")
    assert col.count() == 63
    emb = col._emb
    assert emb.shape == (63, 384) and np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-5

    # encoder vs the fp32 oracle on the same token ids (tolerance: cosine >= 0.999)
    ef = helper.get_embedding_function()
    ids, lens = ef.tokenizer.encode_batch(col.documents)
    from hybrid_rag_system_amd import weights as W

    want = EO.forward(W.generate(ef.cfg, 2026), ids, lens, ef.cfg).numpy()
    cos = (want * emb).sum(1)
    assert cos.min() >= 0.999, cos.min()

    recalls = []
    for q in QUERIES:
        got = helper.hybrid_query(q, 5, "cfg1")
        qe = helper.embed([q]).cpu().numpy()
        # search parity on identical fp32 embeddings: ids and distances bit-exact
        wd, wr = so.topk(qe, emb, 10, "l2")
        raw = helper.query(q, 10, "cfg1")
        assert raw["ids"][0] == [col.ids[r] for r in wr[0]]
        assert raw["distances"][0] == [float(x) for x in wd[0]]
        # fusion parity: the reference's arithmetic on those candidates
        want_rows = T.fuse([col.documents[r] for r in wr[0]], [col.metadatas[r] for r in wr[0]], [col.ids[r] for r in wr[0]], [float(x) for x in wd[0]], q, 5)
        assert got == want_rows
        assert len(got) == 5 and all(set(g) == {"document", "metadata", "id", "semantic_score", "keyword_score", "combined_score", "distance"} for g in got)
        # recall against the all-fp32 pipeline (oracle encoder + oracle search + reference fusion)
        qi, ql = ef.tokenizer.encode_batch([q])
        qo = EO.forward(W.generate(ef.cfg, 2026), qi, ql, ef.cfg).numpy()
        od, orows = so.topk(qo, want, 10, "l2")
        ref_rows = T.fuse([col.documents[r] for r in orows[0]], [col.metadatas[r] for r in orows[0]], [col.ids[r] for r in orows[0]], [float(x) for x in od[0]], q, 5)
        recalls.append(len({g["id"] for g in got} & {r["id"] for r in ref_rows}) / 5)
    print("hybrid top-5 recall vs fp32 pipeline per query:", recalls)
    # north-star bar: hybrid top-k recall >= 0.99 against the all-fp32 pipeline
    assert np.mean(recalls) >= 0.99, recalls


def test_persistence_where_filter_and_bridge(helper):
    from hybrid_rag_system_amd import bridge
    from hybrid_rag_system_amd.store import Client

    fresh = Client(os.environ["HRS_DATA_DIR"])
    col = fresh.get_or_create_collection("cfg1")
    assert col.count() == 63 and [c.name for c in fresh.list_collections()] == ["cfg1"]
    systems = sorted({m.get("system") for m in col.metadatas if m.get("system")})
    assert systems
    sysname = systems[0]
    n_sys = sum(1 for m in col.metadatas if m.get("system") == sysname)
    res = helper.query("container restart", 5, "cfg1", {"system": sysname})
    assert len(res["metadatas"][0]) == min(5, n_sys) and all(m["system"] == sysname for m in res["metadatas"][0])
    none = helper.query("container restart", 5, "cfg1", {"system": "no-such-system"})
    assert none["ids"] == [[]]
    # the compound filters other callers of the reference build (scripts/hybrid-rag.py:358-370,
    # scripts/improved-router.py:463-477): rows and distances equal the oracle's over the kept rows
    from hybrid_rag_system_amd.store import _match

    word = "docker"
    for where in (
        {"$or": [{"system": s} for s in systems[:2]]},
        {"$and": [{"system": sysname}, {"source": {"$contains": ".md"}}]},
        {"title": {"$contains": word}},
        {"chunk_index": {"$gte": 2, "$lt": 6}},
    ):
        keep = np.array([_match(m, where) for m in col.metadatas], np.uint8)
        got = helper.query("container restart", 7, "cfg1", where)
        qe = helper.embed(["container restart"]).cpu().numpy()
        wd, wr = so.topk(qe, col._emb, 7, "l2", mask=keep)
        want_rows = [int(r) for r in wr[0] if r >= 0]
        assert got["ids"][0] == [col.ids[r] for r in want_rows], where
        assert got["distances"][0] == [float(x) for x in wd[0][: len(want_rows)]]
        assert len(want_rows) == min(7, int(keep.sum()))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bridge.main(["query", QUERIES[0], "--hybrid", "--n-results", "3", "--collection", "cfg1"])
    out = json.loads(buf.getvalue())
    assert out["success"] and out["count"] == 3 and [r["index"] for r in out["results"]] == [1, 2, 3]
    assert all(set(r) == {"index", "document", "full_length", "metadata", "combined_score", "semantic_score", "keyword_score"} for r in out["results"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bridge.main(["query", QUERIES[0], "--n-results", "2", "--collection", "cfg1", "--system", sysname])
    out = json.loads(buf.getvalue())
    assert out["success"] and out["count"] == 2 and "similarity" in out["results"][0] and "distance" in out["results"][0]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bridge.main(["stats", "--collection", "cfg1"])
    st = json.loads(buf.getvalue())
    assert st["success"] and st["stats"]["count"] == 63
    batch = helper.hybrid_query_batch(QUERIES, 5, "cfg1")
    assert [r["id"] for r in batch[0]] == [r["id"] for r in helper.hybrid_query(QUERIES[0], 5, "cfg1")]


def test_more_results_than_one_scan_returns(tmp_path):
    """n_results is unbounded in the reference (`-n 100 --hybrid` asks the store for 200): beyond the scan's
    limit of 128 the store concatenates masked passes; the result equals the oracle's top-k."""
    from hybrid_rag_system_amd.store import Client

    rng = np.random.default_rng(5)
    n, d = 1500, 64
    emb = rng.standard_normal((n, d)).astype(np.float32)
    emb[100:140] = emb[7]  # a run of exact ties across the pass boundary
    col = Client(None).get_or_create_collection("big-k")
    col.add(documents=[f"doc {i}" for i in range(n)], metadatas=[{"system": "a" if i % 3 else "b"} for i in range(n)], ids=[f"id{i}" for i in range(n)], embeddings=emb)
    q = rng.standard_normal((2, d)).astype(np.float32)
    q[1] = emb[7]
    for k, where in ((300, None), (129, None), (400, {"system": "b"}), (5000, None), (600, {"system": "b"})):
        res = col.query(query_embeddings=q, n_results=k, where=where)
        keep = np.array([where is None or m["system"] == where["system"] for m in col.metadatas], np.uint8)
        kk = min(k, n)
        wd, wr = so.topk(q, emb, kk, "l2", mask=keep)
        for b in range(2):
            want = [int(r) for r in wr[b] if r >= 0]
            assert len(want) == min(kk, int(keep.sum()))
            assert res["ids"][b] == [f"id{r}" for r in want], (k, where, b)
            assert res["distances"][b] == [float(x) for x in wd[b][: len(want)]]


def test_incremental_collection_and_device_side_where(tmp_path):
    """Rows arrive in many small adds between queries: the device matrix, its shadow and the per-key metadata
    codes behind `where` are extended, never rebuilt, and every query equals the oracle over the rows kept by the
    host-side `_match` -- for the equality forms the device evaluates ($eq/$ne/$in/$nin/$and/$or, nested, missing
    keys, None, unknown values) and for the forms that stay on the host."""
    from hybrid_rag_system_amd.store import Client, _match

    rng = np.random.default_rng(9)
    d, total = 32, 2600
    emb = rng.standard_normal((total, d)).astype(np.float32)
    systems = ["docker", "unraid", "proxmox", None]
    metas = []
    for i in range(total):
        m = {"system": systems[i % 4], "n": i % 7, "flag": bool(i % 2)}
        if i % 5 == 0:
            del m["system"]  # a missing key
        if i % 11 == 0:
            m = None
        metas.append(m)
    col = Client(str(tmp_path)).get_or_create_collection("inc-where")
    q = rng.standard_normal((3, d)).astype(np.float32)
    wheres = [
        {"system": "docker"},
        {"system": {"$ne": "docker"}},
        {"system": {"$in": ["unraid", "proxmox", "nothing"]}},
        {"system": {"$nin": ["unraid"]}, "n": {"$in": [1, 2, 3]}},
        {"$or": [{"system": "unraid"}, {"$and": [{"n": 3}, {"flag": True}]}]},
        {"$and": [{"$or": [{"n": 0}, {"n": 1}]}, {"system": {"$ne": "proxmox"}}], "flag": False},
        {"system": "no-such-system"},
        {"system": None},
        {"$or": []},
        {"n": {"$gte": 2, "$lt": 5}},  # evaluated on the host
    ]
    have = 0
    for step in (1100, 7, 1, 40, 1300, 152):
        ids = [f"id{i}" for i in range(have, have + step)]
        col.add(documents=[f"doc {i}" for i in ids], metadatas=metas[have : have + step], ids=ids, embeddings=emb[have : have + step])
        have += step
        for where in wheres:
            keep = np.array([_match(m, where) for m in col.metadatas], np.uint8)
            res = col.query(query_embeddings=q, n_results=9, where=where)
            wd, wr = so.topk(q, emb[:have], 9, "l2", mask=keep)
            for b in range(3):
                want = [int(r) for r in wr[b] if r >= 0]
                assert res["ids"][b] == [f"id{r}" for r in want], (have, where, b)
                assert res["distances"][b] == [float(x) for x in wd[b][: len(want)]]
    assert col._corpus.n == total and col._dev.shape[0] >= total
    again = Client(str(tmp_path)).get_or_create_collection("inc-where")
    assert again.count() == total and again.query(query_embeddings=q, n_results=4)["ids"] == col.query(query_embeddings=q, n_results=4)["ids"]


def test_full_corpus_hybrid_flag_on_the_text_path(helper, monkeypatch):
    """HRS_HYBRID_FULL=1 / full_corpus=True (an extension; off by default): the reference's fused score with its
    substring keyword test, but the n best over the WHOLE collection -- equal to the reference's own fusion
    arithmetic applied to every document listed nearest first."""
    if helper.get_collection("cfg1").count() == 0:
        helper.index_directory(CORPUS_DIR, ["*.md"], "cfg1")
    col = helper.get_collection("cfg1")
    differs = 0
    for q in QUERIES + ["powershell", "docker compose volumes backup"]:
        qe = helper.embed([q]).cpu().numpy()
        wd, wr = so.topk(qe, col._emb, col.count(), "l2")  # every document, nearest first
        want = T.fuse([col.documents[r] for r in wr[0]], [col.metadatas[r] for r in wr[0]], [col.ids[r] for r in wr[0]], [float(x) for x in wd[0]], q, 5)
        got = helper.hybrid_query(q, 5, "cfg1", full_corpus=True)
        assert got == want, q
        plain = helper.hybrid_query(q, 5, "cfg1")
        differs += [g["id"] for g in got] != [p["id"] for p in plain]
        monkeypatch.setenv("HRS_HYBRID_FULL", "1")
        assert helper.hybrid_query(q, 5, "cfg1") == got
        monkeypatch.delenv("HRS_HYBRID_FULL")
        assert helper.hybrid_query(q, 5, "cfg1") == plain  # the default stays the reference's re-rank
    assert differs >= 1  # somewhere a keyword match lies outside the 2n nearest rows


def test_bm25_keyword_scoring_on_the_text_path(helper, monkeypatch):
    """keyword_scoring="bm25" / HRS_KEYWORD_SCORE=bm25 (an extension; off by default): the re-rank of the 2n nearest
    with the candidates' keyword score from textproc.bm25_keyword_score over the collection's own statistics -- equal
    to that arithmetic applied to the plain query's candidates; the default stays the reference's fraction."""
    from hybrid_rag_system_amd import textproc

    if helper.get_collection("cfg1").count() == 0:
        helper.index_directory(CORPUS_DIR, ["*.md"], "cfg1")
    col = helper.get_collection("cfg1")
    low = [d.lower() for d in col.documents]
    avg = sum(len(d.split()) for d in low) / len(low)
    changed = 0
    for q in QUERIES + ["docker compose volumes backup"]:
        kws = helper.extract_keywords(q)
        res = col.query(query_texts=[q], n_results=10, include=["documents", "metadatas", "distances"])
        df = {k: sum(1 for d in low if k in d) for k in kws}
        scored = []
        for doc, meta, id_, dist in zip(res["documents"][0], res["metadatas"][0], res["ids"][0], res["distances"][0]):
            sem = max(0, 1 - dist)
            kw = textproc.bm25_keyword_score(doc, kws, df, len(low), avg)
            scored.append({"document": doc, "metadata": meta, "id": id_, "semantic_score": sem, "keyword_score": kw, "combined_score": sem * 0.7 + kw * 0.3, "distance": dist})
        scored.sort(key=lambda r: r["combined_score"], reverse=True)
        got = helper.hybrid_query(q, 5, "cfg1", keyword_scoring="bm25")
        assert got == scored[:5], q
        plain = helper.hybrid_query(q, 5, "cfg1")
        changed += [g["keyword_score"] for g in got] != [p["keyword_score"] for p in plain]
        monkeypatch.setenv("HRS_KEYWORD_SCORE", "bm25")
        assert helper.hybrid_query(q, 5, "cfg1") == got
        monkeypatch.delenv("HRS_KEYWORD_SCORE")
        assert helper.hybrid_query(q, 5, "cfg1") == plain
    assert changed >= 1
    qs = QUERIES[:3]
    assert helper.hybrid_query_batch(qs, 5, "cfg1", keyword_scoring="bm25") == [helper.hybrid_query(q, 5, "cfg1", keyword_scoring="bm25") for q in qs]
    with pytest.raises(ValueError):
        helper.hybrid_query("x y z", 5, "cfg1", keyword_scoring="tfidf")
