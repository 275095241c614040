"""Resident engine: one process that keeps the encoder weights and every collection's rows in HBM and
answers the one-shot CLI / bridge invocations over a Unix socket -- the place the Chroma server has in
the reference (scripts/chromadb_helper.py:74-86), so that a `query` pays neither the interpreter's
imports nor the upload of weights and rows.

  python -m hybrid_rag_system_amd.server --socket /run/hrs.sock [--data-dir DIR]
  HRS_SERVER=/run/hrs.sock python scripts/chromadb_helper.py query "..." --hybrid

One request at a time (a context is not thread-safe); many clients may stay connected.  The wire
format and the client are in `remote.py`.  `serve()` takes any object with the `store.Client` surface.
"""
import argparse
import json
import os
import selectors
import signal
import socket
import struct
import sys

from .remote import MAX_MESSAGE


class _Conn:
    def __init__(self, sock):
        self.sock, self.buf = sock, bytearray()

    def messages(self):
        """Complete requests received so far."""
        out = []
        while len(self.buf) >= 4:
            (n,) = struct.unpack(">I", self.buf[:4])
            if n > MAX_MESSAGE:
                raise ConnectionError("oversized request")
            if len(self.buf) < 4 + n:
                break
            out.append(json.loads(self.buf[4 : 4 + n].decode("utf-8")))
            del self.buf[: 4 + n]
        return out


def dispatch(client, req, embed=None):
    """One request against a store.Client-like object -> the JSON-able result."""
    if not isinstance(req, dict):
        raise ValueError("a request is a JSON object")
    op = req.get("op")
    if op == "ping":
        return {"pid": os.getpid(), "collections": [c.name for c in client.list_collections()]}
    if op == "list_collections":
        return [{"name": c.name, "metadata": c.metadata} for c in client.list_collections()]
    if op == "embed":
        if embed is None:
            raise ValueError("this server has no encoder")
        return embed(req["texts"])
    if op == "get_or_create_collection":
        col = client.get_or_create_collection(req["name"], metadata=req.get("metadata"))
        return {"name": col.name, "metadata": col.metadata}
    if op not in ("count", "peek", "get", "add", "query", "query_keyword_classes", "keyword_stats"):
        raise ValueError(f"unknown op {op!r}")
    if not isinstance(req.get("name"), str):
        raise ValueError(f"{op}: collection name missing")
    col = client.get_or_create_collection(req["name"])
    if op == "count":
        return col.count()
    if op == "peek":
        return col.peek(limit=req.get("limit", 10))
    if op == "get":
        return col.get(ids=req.get("ids"))
    if op == "add":
        kw = {"documents": req.get("documents"), "metadatas": req.get("metadatas"), "ids": req.get("ids")}
        if req.get("embeddings") is not None:
            kw["embeddings"] = req["embeddings"]
        col.add(**kw)
        return None
    if op == "keyword_stats":
        return list(col.keyword_stats(req["keywords"]))
    if op == "query_keyword_classes":
        return col.query_keyword_classes(req["query_texts"], req["keywords"], req.get("n_results", 10), req.get("include"))
    kw = {k: req[k] for k in ("query_texts", "n_results", "include", "where", "query_embeddings") if req.get(k) is not None}
    return col.query(**kw)


def serve(client, address, embed=None, ready=None):
    """Accept connections on the Unix socket `address` until a {"op": "shutdown"} request or SIGTERM."""
    if os.path.exists(address):
        os.unlink(address)
    lsock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    old_umask = os.umask(0o177)  # the socket is created 0600: nobody else can connect, not even before the chmod
    try:
        lsock.bind(address)
    finally:
        os.umask(old_umask)
    os.chmod(address, 0o600)
    lsock.listen(64)
    lsock.setblocking(False)
    sel = selectors.DefaultSelector()
    sel.register(lsock, selectors.EVENT_READ, None)
    stop = {"now": False}
    try:
        signal.signal(signal.SIGTERM, lambda *a: stop.update(now=True))
    except ValueError:
        pass  # not the main thread (tests run the loop in a thread)
    if ready is not None:
        ready()

    def reply(sock, obj):
        data = json.dumps(obj).encode("utf-8")
        sock.setblocking(True)
        sock.sendall(struct.pack(">I", len(data)) + data)
        sock.setblocking(False)

    try:
        while not stop["now"]:
            for key, _ in sel.select(timeout=0.5):
                if key.data is None:
                    s, _addr = lsock.accept()
                    s.setblocking(False)
                    sel.register(s, selectors.EVENT_READ, _Conn(s))
                    continue
                conn = key.data
                try:
                    part = conn.sock.recv(1 << 20)
                    if not part:
                        raise ConnectionError("closed")
                    conn.buf += part
                    for req in conn.messages():
                        if isinstance(req, dict) and req.get("op") == "shutdown":
                            reply(conn.sock, {"ok": True, "result": None})
                            stop["now"] = True
                            break
                        try:
                            out = {"ok": True, "result": dispatch(client, req, embed)}
                        except Exception as e:  # reported to the caller, who raises it again
                            out = {"ok": False, "error": str(e), "type": type(e).__name__}
                        reply(conn.sock, out)
                except Exception:  # whatever one client does wrong costs that client its connection, not the server
                    sel.unregister(conn.sock)
                    conn.sock.close()
    finally:
        for key in list(sel.get_map().values()):
            key.fileobj.close()
        sel.close()
        if os.path.exists(address):
            os.unlink(address)


def main(argv=None):
    ap = argparse.ArgumentParser(description="resident hrs engine")
    ap.add_argument("--socket", default=os.getenv("HRS_SERVER"), help="Unix socket path (default $HRS_SERVER)")
    ap.add_argument("--data-dir", default=None, help="collections directory (default $HRS_DATA_DIR)")
    args = ap.parse_args(argv)
    if not args.socket:
        ap.error("--socket or $HRS_SERVER is required")
    os.environ.pop("HRS_SERVER", None)  # this process is the engine, not a client of one
    if args.data_dir:
        os.environ["HRS_DATA_DIR"] = args.data_dir
    from . import helper

    ef = helper.get_embedding_function()  # loads the library and the weights (raises without an MI355X)
    client = helper.get_client()
    resident = []
    for col in client.list_collections():  # rows to HBM before the first request
        col._ef = ef
        if col.count():
            col._resident()
        resident.append(col.name)
    ef.embed_tensor(["warm up"])

    class _WithEncoder:
        """get_or_create_collection that always attaches the process-wide encoder."""

        def get_or_create_collection(self, name, embedding_function=None, metadata=None):
            return client.get_or_create_collection(name, embedding_function=ef, metadata=metadata)

        def list_collections(self):
            return client.list_collections()

    def ready():
        print(json.dumps({"ready": True, "socket": args.socket, "pid": os.getpid(), "collections": resident, "encoder": ef.source}), flush=True)

    serve(_WithEncoder(), args.socket, embed=lambda texts: ef(list(texts)), ready=ready)
    return 0


if __name__ == "__main__":
    sys.exit(main())
