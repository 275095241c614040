"""The resident-engine protocol (server.py / remote.py) on the CPU: a stand-in collection behind the
socket; the helper reached through HRS_SERVER must give what it gives in process."""
import json
import os
import threading

import pytest

from hybrid_rag_system_amd import helper, remote, server


class FakeCollection:
    def __init__(self, name, metadata):
        self.name, self.metadata = name, metadata
        self.ids, self.documents, self.metadatas = [], [], []

    def count(self):
        return len(self.ids)

    def add(self, documents=None, metadatas=None, ids=None, embeddings=None):
        if ids is None:
            raise ValueError("ids are required")
        for d, m, i in zip(documents, metadatas, ids):
            if i not in self.ids:
                self.ids.append(i), self.documents.append(d), self.metadatas.append(m)

    def query(self, query_texts=None, n_results=10, include=None, where=None, query_embeddings=None):
        if query_texts == ["explode"]:
            raise RuntimeError("collection exploded")
        rows = [r for r in range(len(self.ids)) if not where or all(self.metadatas[r].get(k) == v for k, v in where.items())][:n_results]
        one = lambda seq: [[seq[r] for r in rows] for _ in query_texts]
        # awkward floats on purpose: they must cross the wire unchanged
        return {"ids": one(self.ids), "documents": one(self.documents), "metadatas": one(self.metadatas), "distances": [[0.1 + r / 3.0 for r in rows] for _ in query_texts]}

    def peek(self, limit=10):
        return {"ids": self.ids[:limit], "documents": self.documents[:limit], "metadatas": self.metadatas[:limit]}

    def get(self, ids=None):
        return {"ids": list(self.ids)}


class FakeClient:
    def __init__(self):
        self.cols = {}

    def get_or_create_collection(self, name, embedding_function=None, metadata=None):
        if name not in self.cols:
            self.cols[name] = FakeCollection(name, metadata)
        return self.cols[name]

    def list_collections(self):
        return [self.cols[n] for n in sorted(self.cols)]


@pytest.fixture()
def served(tmp_path, monkeypatch):
    backend = FakeClient()
    addr = str(tmp_path / "hrs.sock")
    up = threading.Event()
    th = threading.Thread(target=server.serve, args=(backend, addr), kwargs={"embed": lambda texts: [[float(len(t)), 0.5] for t in texts], "ready": up.set}, daemon=True)
    th.start()
    assert up.wait(10)
    monkeypatch.setenv("HRS_SERVER", addr)
    monkeypatch.setattr(helper, "_client", None)
    yield backend, addr
    try:
        remote.RemoteClient(addr).call("shutdown")
    except ConnectionError:
        pass
    th.join(10)
    assert not th.is_alive() and not os.path.exists(addr)


def test_helper_through_the_socket_equals_in_process(served, tmp_path, monkeypatch):
    backend, addr = served
    f = tmp_path / "guide.md"
    f.write_text("# Title\n" + "docker compose unraid word. " * 120)
    r = helper.index_file(str(f), "docs")
    assert r["status"] == "success" and r["chunks"] == backend.cols["docs"].count() > 1
    assert helper.add_document("proxmox note", {"source": "n.md"}, collection_name="docs")["status"] == "success"
    over_socket = {
        "hybrid": helper.hybrid_query("docker compose", 3, "docs"),
        "plain": helper.query("docker compose", 4, "docs"),
        "where": helper.query("x", 5, "docs", {"system": "proxmox"}),
        "batch": helper.hybrid_query_batch(["docker", "unraid word"], 2, "docs"),
        "stats": helper.get_collection_stats("docs"),
        "list": helper.list_collections(),
        "error": helper.query("explode", 1, "docs"),
    }
    assert over_socket["error"] == {"error": "collection exploded"}
    assert over_socket["list"] == [{"name": "docs", "count": backend.cols["docs"].count(), "metadata": {"description": "Infrastructure and homelab documentation"}}]
    assert len(over_socket["where"]["ids"][0]) == 1
    with pytest.raises(RuntimeError, match="collection exploded"):  # hybrid_query does not catch
        helper.hybrid_query("explode", 1, "docs")
    assert helper.embed(["abc", "de"]) == [[3.0, 0.5], [2.0, 0.5]]
    # the same calls against the same objects without the socket
    monkeypatch.delenv("HRS_SERVER")
    monkeypatch.setattr(helper, "_client", backend)
    monkeypatch.setattr(helper, "get_embedding_function", lambda: None)
    in_process = {
        "hybrid": helper.hybrid_query("docker compose", 3, "docs"),
        "plain": helper.query("docker compose", 4, "docs"),
        "where": helper.query("x", 5, "docs", {"system": "proxmox"}),
        "batch": helper.hybrid_query_batch(["docker", "unraid word"], 2, "docs"),
        "stats": helper.get_collection_stats("docs"),
        "list": helper.list_collections(),
        "error": helper.query("explode", 1, "docs"),
    }
    assert json.dumps(over_socket, sort_keys=True) == json.dumps(in_process, sort_keys=True)
    assert over_socket["hybrid"][0]["distance"] == 0.1 + 0 / 3.0


def test_protocol_edges(served):
    backend, addr = served
    a, b = remote.RemoteClient(addr), remote.RemoteClient(addr)  # two clients stay connected
    assert a.call("ping")["collections"] == [] and b.call("ping")["pid"] == os.getpid()
    with pytest.raises(remote.RemoteError, match="unknown op"):
        a.call("no-such-op")
    with pytest.raises(remote.RemoteError, match="ids are required"):
        b.get_or_create_collection("c").add(documents=["x"], metadatas=[{}])
    big = "x" * (3 << 20)  # larger than one recv
    col = a.get_or_create_collection("d", metadata={"k": 1})
    col.add(documents=[big], metadatas=[{"n": 1}], ids=["id0"])
    assert col.count() == 1 and col.metadata == {"k": 1} and b.get_or_create_collection("d").peek(1)["documents"][0] == big
    # a request that is not a JSON object is answered with an error and the server lives on
    import json as _json
    import socket as _socket
    import struct as _struct

    raw = _socket.socket(_socket.AF_UNIX, _socket.SOCK_STREAM)
    raw.connect(addr)
    body = _json.dumps([]).encode()
    raw.sendall(_struct.pack(">I", len(body)) + body)
    (n,) = _struct.unpack(">I", raw.recv(4))
    ans = _json.loads(raw.recv(n))
    assert ans["ok"] is False and "JSON object" in ans["error"]
    raw.close()
    assert a.call("ping")["pid"] == os.getpid()
    assert oct(os.stat(addr).st_mode & 0o777) == "0o600"
    a.close(), b.close()
    with pytest.raises(ConnectionError, match="no engine server"):
        remote.RemoteClient(addr + ".absent")
