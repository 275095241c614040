"""In-process stand-in for the ChromaDB client + server the reference talks to.

The reference helper calls exactly five things below `get_collection`
(scripts/chromadb_helper.py:74-86, 269, 289-293, 397-401, 539, 544-551, 562-565):
`client.get_or_create_collection`, `client.list_collections`, `collection.add`, `collection.query`,
`collection.count/peek/name/metadata`.  This module offers objects with those methods; the vectors
live in HBM and every distance / top-k comes from libhrs.so.  Persistence is a directory per
collection (fp32 embeddings + json lines), because there is no server to keep state between the
`index` and `query` invocations of the CLI.
"""
import json
import os

import numpy as np
import torch

from . import weights as W
from ._native import MAX_K
from .tokenizer import HashTokenizer, WordPieceTokenizer

_ENGINE = None


def engine():
    """One Engine (hrs_ctx) per process, on the GPU named by HRS_DEVICE (default 0)."""
    global _ENGINE
    if _ENGINE is None:
        from .engine import Engine

        _ENGINE = Engine(int(os.environ.get("HRS_DEVICE", "0")))
    return _ENGINE


class HrsEmbeddingFunction:
    """Callable List[str] -> List[List[float]] like SentenceTransformerEmbeddingFunction
    (scripts/chromadb_helper.py:64-66): tokenise on the host, encode on the GPU, unit-norm rows.

    model_dir (or $HRS_MODEL_DIR) may hold `vocab.txt` + `model.safetensors` of a MiniLM-L6 class
    model; without it the generated weights and the hash tokeniser are used (no hub access exists).
    """

    def __init__(self, model_name="all-MiniLM-L6-v2", model_dir=None, seed=2026, batch_size=1024):
        self.model_name = model_name
        self.cfg = W.EncoderConfig()
        model_dir = model_dir or os.environ.get("HRS_MODEL_DIR")
        if model_dir and os.path.exists(os.path.join(model_dir, "model.safetensors")):
            weights = W.load_safetensors(model_dir, self.cfg)
            self.tokenizer = WordPieceTokenizer(os.path.join(model_dir, "vocab.txt"))
            self.source = model_dir
        else:
            weights = W.generate(self.cfg, seed)
            self.tokenizer = HashTokenizer(self.cfg.vocab)
            self.source = f"generated(seed={seed})"
        self.encoder = engine().encoder(weights, self.cfg)
        self.batch_size = batch_size

    def embed_tensor(self, texts):
        """-> fp32 [len(texts), 384] CUDA tensor."""
        eng = engine()
        n = len(texts)
        if n == 0:
            return torch.empty((0, self.cfg.hidden), dtype=torch.float32, device=eng.device)
        # length-sorted batches keep the padding small (sentence-transformers sorts by length too)
        order = sorted(range(n), key=lambda i: len(texts[i])) if n > 1 else [0]
        parts = []
        for s in range(0, n, self.batch_size):
            ids, lens = self.tokenizer.encode_batch([texts[i] for i in order[s : s + self.batch_size]])
            parts.append(self.encoder.encode_ids(torch.from_numpy(ids).to(eng.device), torch.from_numpy(lens).to(eng.device)))
        emb = parts[0] if len(parts) == 1 else torch.cat(parts)
        if n == 1:
            return emb
        back = np.empty(n, np.int64)
        back[np.asarray(order)] = np.arange(n)  # one gather restores the caller's order
        return emb[torch.from_numpy(back).to(eng.device)]

    def __call__(self, input):
        return self.embed_tensor(list(input)).cpu().numpy().tolist()


_COMPARE = {"$gt": lambda v, a: v > a, "$gte": lambda v, a: v >= a, "$lt": lambda v, a: v < a, "$lte": lambda v, a: v <= a}


def _match(meta, where):
    """ChromaDB `where` subset: {"k": v}, {"k": {"$eq"|"$ne"|"$in"|"$nin"|"$gt"|"$gte"|"$lt"|"$lte"|"$contains": ...}},
    {"$and": [...]}, {"$or": [...]} -- the forms the reference's callers build (scripts/chromadb_helper.py:265-266,
    scripts/hybrid-rag.py:358-370, scripts/improved-router.py:463-477).  A missing key matches only `$ne` / `$nin`."""
    for key, cond in where.items():
        if key == "$and":
            if not all(_match(meta, w) for w in cond):
                return False
        elif key == "$or":
            if not any(_match(meta, w) for w in cond):
                return False
        elif isinstance(cond, dict):
            v = (meta or {}).get(key)
            for op, arg in cond.items():
                if op == "$eq":
                    ok = v == arg
                elif op == "$ne":
                    ok = v != arg
                elif op == "$in":
                    ok = v in arg
                elif op == "$nin":
                    ok = v not in arg
                elif op in _COMPARE:
                    ok = isinstance(v, (int, float)) and not isinstance(v, bool) and isinstance(arg, (int, float)) and _COMPARE[op](v, arg)
                elif op == "$contains":  # substring of a string value, or membership in a list value
                    ok = isinstance(v, (str, list, tuple)) and arg in v
                else:
                    raise ValueError(f"unsupported where operator {op}")
                if not ok:
                    return False
        elif (meta or {}).get(key) != cond:
            return False
    return True


class Collection:
    def __init__(self, client, name, embedding_function=None, metadata=None, space="l2"):
        self._client = client
        self.name = name
        self.metadata = metadata
        self._ef = embedding_function
        self.space = space
        self.ids, self.documents, self.metadatas = [], [], []
        self._index = {}
        self._emb = np.zeros((0, 0), np.float32)
        self._corpus = None
        self._dev = None
        self._load()

    # ---- persistence ---------------------------------------------------------------------------
    def _dir(self):
        return os.path.join(self._client.path, self.name) if self._client.path else None

    def _load(self):
        d = self._dir()
        if not d or not os.path.exists(os.path.join(d, "meta.json")):
            return
        with open(os.path.join(d, "meta.json")) as f:
            meta = json.load(f)
        self.metadata = meta.get("metadata", self.metadata)
        self.space = meta.get("space", self.space)
        with open(os.path.join(d, "rows.jsonl")) as f:
            for line in f:
                r = json.loads(line)
                self.ids.append(r["id"])
                self.documents.append(r["document"])
                self.metadatas.append(r["metadata"])
        self._index = {i: n for n, i in enumerate(self.ids)}
        dim = meta["dim"]
        self._emb = np.fromfile(os.path.join(d, "embeddings.f32"), np.float32).reshape(-1, dim) if dim else self._emb
        if len(self.ids) != self._emb.shape[0]:
            raise RuntimeError(f"collection {self.name}: {len(self.ids)} rows but {self._emb.shape[0]} embeddings on disk")

    def _append_to_disk(self, first):
        d = self._dir()
        if not d:
            return
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "rows.jsonl"), "a") as f:
            for n in range(first, len(self.ids)):
                f.write(json.dumps({"id": self.ids[n], "document": self.documents[n], "metadata": self.metadatas[n]}) + "\n")
        with open(os.path.join(d, "embeddings.f32"), "ab") as f:
            self._emb[first:].tofile(f)
        with open(os.path.join(d, "meta.json"), "w") as f:
            json.dump({"name": self.name, "metadata": self.metadata, "space": self.space, "dim": int(self._emb.shape[1]), "count": len(self.ids)}, f)

    # ---- the five methods the helper uses ------------------------------------------------------
    def count(self):
        return len(self.ids)

    def add(self, documents=None, metadatas=None, ids=None, embeddings=None):
        if ids is None:
            raise ValueError("ids are required")
        n = len(ids)
        documents = list(documents) if documents is not None else [None] * n
        metadatas = list(metadatas) if metadatas is not None else [None] * n
        if len(documents) != n or len(metadatas) != n:
            raise ValueError("documents, metadatas and ids must have the same length")
        if len(set(ids)) != n:
            raise ValueError("duplicate ids in one add() call")
        fresh = [j for j in range(n) if ids[j] not in self._index]  # existing ids are kept, not overwritten
        if not fresh:
            return
        if embeddings is None:
            if self._ef is None:
                raise ValueError("no embeddings and no embedding function")
            emb = self._ef.embed_tensor([documents[j] for j in fresh]).cpu().numpy() if hasattr(self._ef, "embed_tensor") else np.asarray(self._ef([documents[j] for j in fresh]), np.float32)
        else:
            emb = np.asarray(embeddings, np.float32)[fresh]
        if self._emb.shape[0] == 0:
            self._emb = np.zeros((0, emb.shape[1]), np.float32)
        if emb.shape[1] != self._emb.shape[1]:
            raise ValueError(f"embedding dimension {emb.shape[1]} != collection dimension {self._emb.shape[1]}")
        first = len(self.ids)
        for j in fresh:
            self._index[ids[j]] = len(self.ids)
            self.ids.append(ids[j])
            self.documents.append(documents[j])
            self.metadatas.append(metadatas[j])
        # rows live in a buffer that doubles, so a long run of small adds stays linear
        need = first + emb.shape[0]
        buf = getattr(self, "_buf", None)
        if buf is None or buf.shape[1] != emb.shape[1] or buf.shape[0] < need or buf[:first].base is not self._emb.base and not np.shares_memory(buf, self._emb):
            cap = max(need, 2 * first, 1024)
            nb = np.empty((cap, emb.shape[1]), np.float32)
            nb[:first] = self._emb
            self._buf = buf = nb
        buf[first:need] = emb
        self._emb = buf[:need]
        self._corpus = None
        self._append_to_disk(first)

    def _resident(self):
        if self._corpus is None:
            eng = engine()
            self._dev = torch.from_numpy(self._emb).to(eng.device).contiguous()
            self._corpus = eng.corpus(self._dev)
        return self._corpus

    def query(self, query_texts=None, n_results=10, include=None, where=None, query_embeddings=None):
        include = list(include) if include is not None else ["documents", "metadatas", "distances"]
        eng = engine()
        if query_embeddings is None:
            if query_texts is None:
                raise ValueError("query_texts or query_embeddings required")
            q = self._ef.embed_tensor(list(query_texts)) if hasattr(self._ef, "embed_tensor") else torch.tensor(self._ef(list(query_texts)), dtype=torch.float32, device=eng.device)
        else:
            q = torch.as_tensor(np.asarray(query_embeddings, np.float32)).to(eng.device)
        nq = q.shape[0]
        rows = dist = None
        if self.count() and n_results > 0:
            k = min(int(n_results), self.count())
            mask = None
            if where:
                keep = np.fromiter((_match(m, where) for m in self.metadatas), np.uint8, len(self.metadatas))
                mask = torch.from_numpy(keep).to(eng.device)
            if k <= MAX_K:
                d, r = eng.score_topk(self._resident(), q.contiguous(), k, space=self.space, mask=mask)
            else:
                d, r = self._topk_beyond_max(q.contiguous(), k, mask)
            rows, dist = r.cpu().numpy(), d.cpu().numpy()
        res = {"ids": [], "documents": [] if "documents" in include else None, "metadatas": [] if "metadatas" in include else None, "distances": [] if "distances" in include else None}
        for b in range(nq):
            rr = [int(x) for x in rows[b] if x >= 0] if rows is not None else []
            res["ids"].append([self.ids[x] for x in rr])
            if res["documents"] is not None:
                res["documents"].append([self.documents[x] for x in rr])
            if res["metadatas"] is not None:
                res["metadatas"].append([self.metadatas[x] for x in rr])
            if res["distances"] is not None:
                res["distances"].append([float(dist[b][j]) for j in range(len(rr))])
        return res

    def _topk_beyond_max(self, q, k, mask):
        """More results than one scan returns (n_results is unbounded in the reference): per query, passes of
        MAX_K with the rows already returned masked out.  The order is total (distance, then row), so the
        passes concatenate to the exact top-k."""
        eng = engine()
        n = self.count()
        dist = torch.full((q.shape[0], k), float("inf"), dtype=torch.float32, device=eng.device)
        rows = torch.full((q.shape[0], k), -1, dtype=torch.int64, device=eng.device)
        for b in range(q.shape[0]):
            keep = mask.clone() if mask is not None else torch.ones(n, dtype=torch.uint8, device=eng.device)
            got = 0
            while got < k:
                step = min(MAX_K, k - got)
                d, r = eng.score_topk(self._resident(), q[b : b + 1], step, space=self.space, mask=keep)
                found = int((r[0] >= 0).sum())
                dist[b, got : got + found] = d[0, :found]
                rows[b, got : got + found] = r[0, :found]
                got += found
                if found < step:
                    break
                keep[r[0, :found]] = 0
        return dist, rows

    def peek(self, limit=10):
        n = min(limit, self.count())
        return {"ids": self.ids[:n], "documents": self.documents[:n], "metadatas": self.metadatas[:n], "embeddings": self._emb[:n].tolist()}

    def get(self, ids=None):
        idx = range(self.count()) if ids is None else [self._index[i] for i in ids if i in self._index]
        return {"ids": [self.ids[i] for i in idx], "documents": [self.documents[i] for i in idx], "metadatas": [self.metadatas[i] for i in idx]}


class Client:
    """Stands in for chromadb.HttpClient(host, port) (scripts/chromadb_helper.py:74-76)."""

    def __init__(self, path=None):
        self.path = path
        self._open = {}

    def get_or_create_collection(self, name, embedding_function=None, metadata=None):
        if name not in self._open:
            self._open[name] = Collection(self, name, embedding_function, metadata)
        col = self._open[name]
        if embedding_function is not None:
            col._ef = embedding_function
        return col

    def list_collections(self):
        names = set(self._open)
        if self.path and os.path.isdir(self.path):
            names |= {d for d in os.listdir(self.path) if os.path.exists(os.path.join(self.path, d, "meta.json"))}
        return [self.get_or_create_collection(n) for n in sorted(names)]
