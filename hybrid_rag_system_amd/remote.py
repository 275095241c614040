"""Client side of the resident engine (`server.py`): the same five collection methods as `store.Client`,
forwarded over a Unix socket, so the one-shot CLI and the bridge answer without loading torch, the
library or the weights.  It takes the place of `chromadb.HttpClient(host, port)` in the reference
(scripts/chromadb_helper.py:74-76); texts are embedded on the server's GPU.

Standard library only.  Wire format: a 4-byte big-endian length, then that many bytes of UTF-8 JSON;
request {"op": str, ...}, reply {"ok": true, "result": ...} or {"ok": false, "error": str}.  Python
floats round-trip through JSON exactly, so results equal those of the in-process store bit for bit.
"""
import json
import socket
import struct

MAX_MESSAGE = 1 << 30


def send_msg(sock, obj):
    data = json.dumps(obj).encode("utf-8")
    sock.sendall(struct.pack(">I", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(min(n - len(buf), 1 << 20))
        if not part:
            raise ConnectionError("connection closed by the engine server")
        buf += part
    return bytes(buf)


def recv_msg(sock):
    (n,) = struct.unpack(">I", _recv_exact(sock, 4))
    if n > MAX_MESSAGE:
        raise ConnectionError(f"message of {n} bytes refused")
    return json.loads(_recv_exact(sock, n).decode("utf-8"))


class RemoteError(RuntimeError):
    """An exception raised on the server; str() is the server-side message, as the helper reports it."""


class RemoteClient:
    def __init__(self, address, timeout=60.0):
        self.address = address
        self._sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self._sock.settimeout(timeout)
        try:
            self._sock.connect(address)
        except OSError as e:
            self._sock.close()
            raise ConnectionError(f"no engine server at {address}: {e}") from e

    def call(self, op, **kw):
        send_msg(self._sock, dict(kw, op=op))
        reply = recv_msg(self._sock)
        if not reply.get("ok"):
            raise RemoteError(reply.get("error", "unknown server error"))
        return reply.get("result")

    def close(self):
        self._sock.close()

    # ---- the store.Client surface ---------------------------------------------------------------
    def get_or_create_collection(self, name, embedding_function=None, metadata=None):
        info = self.call("get_or_create_collection", name=name, metadata=metadata)
        return RemoteCollection(self, info["name"], info["metadata"])

    def list_collections(self):
        return [RemoteCollection(self, c["name"], c["metadata"]) for c in self.call("list_collections")]

    def embed(self, texts):
        return self.call("embed", texts=list(texts))


class RemoteCollection:
    _ef = None  # texts are embedded on the server

    def __init__(self, client, name, metadata):
        self._client, self.name, self.metadata = client, name, metadata

    def count(self):
        return self._client.call("count", name=self.name)

    def peek(self, limit=10):
        return self._client.call("peek", name=self.name, limit=limit)

    def get(self, ids=None):
        return self._client.call("get", name=self.name, ids=ids)

    def add(self, documents=None, metadatas=None, ids=None, embeddings=None):
        if embeddings is not None and hasattr(embeddings, "tolist"):
            embeddings = embeddings.tolist()
        self._client.call("add", name=self.name, documents=documents, metadatas=metadatas, ids=ids, embeddings=embeddings)

    def query(self, query_texts=None, n_results=10, include=None, where=None, query_embeddings=None):
        if query_embeddings is not None and hasattr(query_embeddings, "tolist"):
            query_embeddings = query_embeddings.tolist()
        return self._client.call(
            "query", name=self.name, query_texts=query_texts, n_results=n_results, include=include, where=where, query_embeddings=query_embeddings
        )

    def query_keyword_classes(self, query_texts, keywords, n_results=10, include=None):
        return self._client.call("query_keyword_classes", name=self.name, query_texts=list(query_texts), keywords=[list(k) for k in keywords], n_results=n_results, include=include)

    def keyword_stats(self, keywords):
        df, n, avg = self._client.call("keyword_stats", name=self.name, keywords=list(keywords))
        return df, n, avg
