"""The on-disk collection (the stand-in for the ChromaDB server's volume): concurrent writers, a writer that
died mid-append, collection names.  No GPU: rows are added with explicit embeddings."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from hybrid_rag_system_amd.store import Client, validate_name

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WRITER = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from hybrid_rag_system_amd.store import Client
who, path = int(sys.argv[1]), sys.argv[2]
for batch in range(40):
    col = Client(path).get_or_create_collection("shared")       # a fresh process-like view per batch, as the CLI has
    ids = [f"w{{who}}-{{batch}}-{{j}}" for j in range(5)] + [f"common-{{batch}}"]
    emb = np.array([[float(sum(map(ord, i))), float(len(i)), float(who), 0.0] for i in ids], np.float32)
    col.add(documents=[f"doc of {{i}}" for i in ids], metadatas=[{{"who": who}} for _ in ids], ids=ids, embeddings=emb)
"""


def test_two_concurrent_writers_keep_rows_and_vectors_aligned(tmp_path):
    path = str(tmp_path / "data")
    script = tmp_path / "writer.py"
    script.write_text(WRITER.format(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(script), str(w), path]) for w in (1, 2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    col = Client(path).get_or_create_collection("shared")
    assert col.count() == 2 * 40 * 5 + 40  # the common ids were added once, by whoever came first
    assert len(set(col.ids)) == col.count()
    for i, doc, meta, e in zip(col.ids, col.documents, col.metadatas, col._emb):
        assert doc == f"doc of {i}"
        assert e[0] == float(sum(map(ord, i))) and e[1] == float(len(i))  # the vector belongs to the id of its row
        assert e[2] == float(meta["who"])
    meta = json.load(open(os.path.join(path, "shared", "meta.json")))
    assert meta["count"] == col.count() and meta["rows_bytes"] == os.path.getsize(os.path.join(path, "shared", "rows.jsonl"))


def test_tail_of_a_dead_writer_is_cut_off(tmp_path):
    path = str(tmp_path / "data")
    col = Client(path).get_or_create_collection("crash")
    col.add(documents=["a", "b"], metadatas=[{}, {}], ids=["a", "b"], embeddings=np.eye(2, 4, dtype=np.float32))
    d = os.path.join(path, "crash")
    # a writer appended rows and half a vector, then died before replacing meta.json
    with open(os.path.join(d, "rows.jsonl"), "ab") as f:
        f.write(b'{"id": "ghost", "document": "x", "metadata": {}}\n{"id": "torn')
    with open(os.path.join(d, "embeddings.f32"), "ab") as f:
        f.write(b"\x00" * 23)
    again = Client(path).get_or_create_collection("crash")
    assert again.ids == ["a", "b"] and again._emb.shape == (2, 4)
    again.add(documents=["c"], metadatas=[None], ids=["c"], embeddings=np.full((1, 4), 7, np.float32))
    final = Client(path).get_or_create_collection("crash")
    assert final.ids == ["a", "b", "c"] and (final._emb[2] == 7).all() and np.array_equal(final._emb[:2], np.eye(2, 4, dtype=np.float32))
    # committed data missing is corruption, not something to paper over
    with open(os.path.join(d, "embeddings.f32"), "r+b") as f:
        f.truncate(8)
    with pytest.raises(RuntimeError):
        Client(path).get_or_create_collection("crash")


def test_a_writer_that_died_during_the_very_first_add_leaves_nothing_behind(tmp_path):
    """No meta.json yet, but both data files already hold the dead writer's bytes: the next writer must not append
    behind them (its commit record measures rows.jsonl from offset 0)."""
    path = str(tmp_path / "data")
    d = os.path.join(path, "first")
    os.makedirs(d)
    with open(os.path.join(d, "rows.jsonl"), "wb") as f:
        f.write(b'{"id": "ghost", "document": "x", "metadata": {}}\n{"id": "torn')
    with open(os.path.join(d, "embeddings.f32"), "wb") as f:
        f.write(b"\x01" * 37)
    col = Client(path).get_or_create_collection("first")
    assert col.count() == 0
    col.add(documents=["a", "b"], metadatas=[{}, {"k": 1}], ids=["a", "b"], embeddings=np.eye(2, 4, dtype=np.float32))
    again = Client(path).get_or_create_collection("first")
    assert again.ids == ["a", "b"] and again.documents == ["a", "b"]
    assert np.array_equal(again._emb, np.eye(2, 4, dtype=np.float32))
    assert os.path.getsize(os.path.join(d, "embeddings.f32")) == 2 * 4 * 4


def test_an_open_collection_sees_rows_committed_by_others_before_it_adds(tmp_path):
    path = str(tmp_path / "data")
    a = Client(path).get_or_create_collection("two-views")
    a.add(documents=["1"], metadatas=[{}], ids=["1"], embeddings=np.ones((1, 4), np.float32))
    b = Client(path).get_or_create_collection("two-views")
    b.add(documents=["2"], metadatas=[{}], ids=["2"], embeddings=2 * np.ones((1, 4), np.float32))
    a.add(documents=["2 again", "3"], metadatas=[{}, {}], ids=["2", "3"], embeddings=3 * np.ones((2, 4), np.float32))
    assert a.ids == ["1", "2", "3"] and a.documents[1] == "2"  # the id b committed is kept, not overwritten
    assert [float(x) for x in a._emb[:, 0]] == [1.0, 2.0, 3.0]
    assert Client(path).get_or_create_collection("two-views").ids == ["1", "2", "3"]


@pytest.mark.parametrize("name", ["../../x", "a", "ab", "a/b", "x..y", ".hidden", "trailing-", "x" * 64, "", None, "sp ace"])
def test_collection_names_that_could_leave_the_data_directory_are_refused(tmp_path, name):
    with pytest.raises(ValueError):
        Client(str(tmp_path)).get_or_create_collection(name)
    assert not os.path.exists(tmp_path.parent / "x")


def test_valid_names():
    for name in ("infrastructure_docs", "homelab_knowledge", "cfg1", "a.b-c_d", "x" * 63):
        assert validate_name(name) == name


def test_keyword_stats_follow_the_collection(tmp_path):
    """Collection.keyword_stats (behind the BM25 extension): document frequency by substring of the lower-cased
    text, mean length in words, refreshed when documents are added."""
    from hybrid_rag_system_amd import store

    col = store.Collection.__new__(store.Collection)
    col.documents = ["The Proxmox host", "proxmox PROXMOX backup", "nothing"]
    col.count = lambda: len(col.documents)
    df, n, avg = col.keyword_stats(["proxmox", "backup", "zfs"])
    assert (df, n) == ({"proxmox": 2, "backup": 1, "zfs": 0}, 3) and avg == (3 + 3 + 1) / 3
    col.documents.append("zfs pool on proxmox")
    df, n, avg = col.keyword_stats(["proxmox", "zfs"])
    assert (df, n) == ({"proxmox": 3, "zfs": 1}, 4) and avg == (3 + 3 + 1 + 4) / 4
