"""In-process stand-in for the ChromaDB client + server the reference talks to.

The reference helper calls exactly five things below `get_collection`
(scripts/chromadb_helper.py:74-86, 269, 289-293, 397-401, 539, 544-551, 562-565):
`client.get_or_create_collection`, `client.list_collections`, `collection.add`, `collection.query`,
`collection.count/peek/name/metadata`.  This module offers objects with those methods; the vectors
live in HBM and every distance / top-k comes from libhrs.so.  Persistence is a directory per
collection (fp32 embeddings + json lines), because there is no server to keep state between the
`index` and `query` invocations of the CLI.
"""
import contextlib
import fcntl
import json
import os
import re

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


_NAME_RE = re.compile(r"^[A-Za-z0-9][A-Za-z0-9._-]{1,61}[A-Za-z0-9]$")


def validate_name(name):
    """ChromaDB's collection-name rule: 3-63 characters from [A-Za-z0-9._-], first and last alphanumeric, no "..".
    The name becomes a directory under the data directory, so nothing else may pass (no separators, no "..")."""
    if not isinstance(name, str) or not _NAME_RE.match(name) or ".." in name:
        raise ValueError(f"invalid collection name {name!r}: expected 3-63 characters from [A-Za-z0-9._-], starting and ending with a letter or digit, without '..'")
    return name


_DEVICE_OPS = ("$eq", "$ne", "$in", "$nin")


class Collection:
    """One collection: ids / documents / metadatas on the host, embeddings on the host (for persistence and peek)
    and in HBM (for every query).  Appends are incremental end to end: the files grow by the new rows, the device
    matrix and its scan shadow grow geometrically (hrs_corpus_append quantises only the new rows), the per-key
    metadata codes behind `where` masks are extended, never rebuilt."""

    def __init__(self, client, name, embedding_function=None, metadata=None, space="l2"):
        self._client = client
        self.name = name
        self.metadata = metadata
        self._ef = embedding_function
        self.space = space
        self.ids, self.documents, self.metadatas = [], [], []
        self._index = {}
        self._buf = np.zeros((0, 0), np.float32)  # host rows, capacity doubles; self._emb is the filled part
        self._emb = self._buf
        self._rows_bytes = 0   # committed length of rows.jsonl that has been read / written by this object
        self._corpus = None    # engine.Corpus over self._dev[: self._dev_rows]
        self._dev = None       # device rows, capacity doubles
        self._dev_rows = 0
        self._codes = {}       # metadata key -> {"map": value -> code, "host": [code per row], "dev": int32 tensor, "n": rows on the device}
        with self._locked():
            self._sync_from_disk()

    # ---- persistence ---------------------------------------------------------------------------
    # Layout: <dir>/rows.jsonl (one {"id", "document", "metadata"} per line), <dir>/embeddings.f32 (row-major
    # fp32), <dir>/meta.json (the COMMIT record: count, dim, rows_bytes), <dir>/.lock (flock).  A writer appends
    # to the two data files, fsyncs, then replaces meta.json atomically; whatever lies beyond the committed
    # lengths is the tail of a writer that died and is cut off by the next one.  The reference leaves all of this
    # to the ChromaDB server, which serialises its writers (scripts/chromadb_helper.py:74-86).
    def _dir(self):
        return os.path.join(self._client.path, self.name) if self._client.path else None

    @contextlib.contextmanager
    def _locked(self):
        d = self._dir()
        if not d or not os.path.isdir(d):  # nothing on disk yet: nothing to guard (add() creates the directory first)
            yield
            return
        fd = os.open(os.path.join(d, ".lock"), os.O_CREAT | os.O_RDWR, 0o600)
        try:
            fcntl.flock(fd, fcntl.LOCK_EX)
            yield
        finally:
            fcntl.flock(fd, fcntl.LOCK_UN)
            os.close(fd)

    def _read_meta(self):
        d = self._dir()
        path = os.path.join(d, "meta.json") if d else None
        if not path or not os.path.exists(path):
            return None
        with open(path) as f:
            return json.load(f)

    def _sync_from_disk(self):
        """Under the lock: take in the rows other processes have committed since this object last looked, and cut
        off any uncommitted tail."""
        meta = self._read_meta()
        d = self._dir()
        if meta is None:
            # Nothing has ever been committed: whatever the data files hold is the tail of a writer that died
            # during the very first add (it had no commit record to be measured against).  Cut it off, or the next
            # writer's rows would land behind it and be committed with lengths counted from offset 0.
            if d and os.path.isdir(d) and not self.ids:
                for name in ("rows.jsonl", "embeddings.f32"):
                    path = os.path.join(d, name)
                    if os.path.exists(path) and os.path.getsize(path) > 0:
                        with open(path, "r+b") as f:
                            f.truncate(0)
            return
        if not self.ids:
            self.metadata = meta.get("metadata", self.metadata)
            self.space = meta.get("space", self.space)
        count, dim = int(meta["count"]), int(meta["dim"])
        rows_path, emb_path = os.path.join(d, "rows.jsonl"), os.path.join(d, "embeddings.f32")
        have = len(self.ids)
        if count < have:
            raise RuntimeError(f"collection {self.name}: {count} rows committed on disk but {have} loaded (files replaced underneath?)")
        rows_bytes = meta.get("rows_bytes")
        with open(rows_path, "rb") as f:
            f.seek(self._rows_bytes)
            if rows_bytes is None:  # written before the commit record held it: count lines
                lines = []
                for _ in range(count - have):
                    line = f.readline()
                    if not line.endswith(b"\n"):
                        break
                    lines.append(line)
                rows_bytes = self._rows_bytes + sum(len(x) for x in lines)
            else:
                blob = f.read(int(rows_bytes) - self._rows_bytes)
                lines = blob.splitlines(keepends=True)
        if len(lines) != count - have or (lines and not lines[-1].endswith(b"\n")):
            raise RuntimeError(f"collection {self.name}: rows.jsonl holds {have + len(lines)} of the {count} committed rows")
        emb_bytes = count * dim * 4
        if os.path.getsize(emb_path) < emb_bytes:
            raise RuntimeError(f"collection {self.name}: embeddings.f32 holds fewer than the {count} committed rows")
        for path, size in ((rows_path, int(rows_bytes)), (emb_path, emb_bytes)):
            if os.path.getsize(path) > size:  # the tail of a writer that died before its commit
                with open(path, "r+b") as f:
                    f.truncate(size)
        if count > have:
            new = np.fromfile(emb_path, np.float32, count=(count - have) * dim, offset=have * dim * 4).reshape(-1, dim)
            recs = [json.loads(x) for x in lines]
            self._take_rows([r["id"] for r in recs], [r["document"] for r in recs], [r["metadata"] for r in recs], new)
        self._rows_bytes = int(rows_bytes)

    def _take_rows(self, ids, documents, metadatas, emb):
        """Append rows to the host-side state (not to disk, not yet to the device)."""
        first = len(self.ids)
        if self._buf.shape[1] != emb.shape[1]:
            if first:
                raise ValueError(f"embedding dimension {emb.shape[1]} != collection dimension {self._buf.shape[1]}")
            self._buf = np.empty((0, emb.shape[1]), np.float32)
        need = first + emb.shape[0]
        if self._buf.shape[0] < need:  # rows live in a buffer that doubles, so a long run of small adds stays linear
            nb = np.empty((max(need, 2 * self._buf.shape[0], 1024), emb.shape[1]), np.float32)
            nb[:first] = self._buf[:first]
            self._buf = nb
        self._buf[first:need] = emb
        self._emb = self._buf[:need]
        for i, doc, meta in zip(ids, documents, metadatas):
            self._index[i] = len(self.ids)
            self.ids.append(i)
            self.documents.append(doc)
            self.metadatas.append(meta)
        for key, st in self._codes.items():  # extend the cached where-codes by the new rows
            st["host"].extend(self._code_of(st, (m or {}).get(key)) for m in metadatas)

    def _append_to_disk(self, first):
        d = self._dir()
        if not d:
            return
        with open(os.path.join(d, "rows.jsonl"), "ab") as f:
            blob = "".join(json.dumps({"id": self.ids[n], "document": self.documents[n], "metadata": self.metadatas[n]}) + "\n" for n in range(first, len(self.ids))).encode("utf-8")
            f.write(blob)
            f.flush()
            os.fsync(f.fileno())
        with open(os.path.join(d, "embeddings.f32"), "ab") as f:
            self._emb[first:].tofile(f)
            f.flush()
            os.fsync(f.fileno())
        self._rows_bytes += len(blob)
        tmp = os.path.join(d, "meta.json.tmp")
        with open(tmp, "w") as f:
            json.dump({"name": self.name, "metadata": self.metadata, "space": self.space, "dim": int(self._emb.shape[1]), "count": len(self.ids), "rows_bytes": self._rows_bytes}, f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, os.path.join(d, "meta.json"))  # the commit

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
        d = self._dir()
        if d:
            os.makedirs(d, exist_ok=True)
        with self._locked():
            self._sync_from_disk()  # rows another process committed meanwhile: their ids count as existing
            fresh = [j for j in range(n) if ids[j] not in self._index]  # existing ids are kept, not overwritten
            if not fresh:
                return
            if embeddings is None:
                if self._ef is None:
                    raise ValueError("no embeddings and no embedding function")
                docs = [documents[j] for j in fresh]
                emb = self._ef.embed_tensor(docs).cpu().numpy() if hasattr(self._ef, "embed_tensor") else np.asarray(self._ef(docs), np.float32)
            else:
                emb = np.asarray(embeddings, np.float32)[fresh]
            if self.ids and emb.shape[1] != self._emb.shape[1]:
                raise ValueError(f"embedding dimension {emb.shape[1]} != collection dimension {self._emb.shape[1]}")
            first = len(self.ids)
            self._take_rows([ids[j] for j in fresh], [documents[j] for j in fresh], [metadatas[j] for j in fresh], emb)
            self._append_to_disk(first)

    def _resident(self):
        """The corpus handle over all rows: uploads and quantises only the rows added since the last call."""
        eng = engine()
        n, dim = self._emb.shape
        if self._corpus is not None and self._dev_rows == n:
            return self._corpus
        if self._dev is None or self._dev.shape[0] < n or self._dev.shape[1] != dim:
            nd = torch.empty((max(n, 2 * (self._dev.shape[0] if self._dev is not None else 0), 1024), dim), dtype=torch.float32, device=eng.device)
            if self._dev is not None and self._dev_rows:
                nd[: self._dev_rows] = self._dev[: self._dev_rows]
            self._dev = nd
        self._dev[self._dev_rows : n] = torch.from_numpy(self._emb[self._dev_rows : n]).to(eng.device)
        if self._corpus is None:
            self._corpus = eng.corpus(self._dev, n=n)
        else:
            self._corpus.append(self._dev, n)
        self._dev_rows = n
        return self._corpus

    # ---- `where` -------------------------------------------------------------------------------
    @staticmethod
    def _code_of(st, value):
        try:
            return st["map"].setdefault(value, len(st["map"]))
        except TypeError:  # an unhashable value (a list): never equal to a scalar operand
            return -1

    def _key_codes(self, key):
        """int32 device tensor [count]: the code of every row's value under `key` (grown, not rebuilt, by adds)."""
        eng = engine()
        st = self._codes.get(key)
        if st is None:
            st = self._codes[key] = {"map": {}, "host": [], "dev": None, "n": 0}
            st["host"] = [self._code_of(st, (m or {}).get(key)) for m in self.metadatas]
        n = len(self.ids)
        if st["n"] < n:
            if st["dev"] is None or st["dev"].numel() < n:
                nd = torch.empty(max(n, 2 * (st["dev"].numel() if st["dev"] is not None else 0), 1024), dtype=torch.int32, device=eng.device)
                if st["n"]:
                    nd[: st["n"]] = st["dev"][: st["n"]]
                st["dev"] = nd
            st["dev"][st["n"] : n] = torch.tensor(st["host"][st["n"] : n], dtype=torch.int32).to(eng.device)
            st["n"] = n
        return st

    @staticmethod
    def _device_filter(where):
        """True when `where` uses only the forms the device evaluates: equality, $eq/$ne/$in/$nin, $and/$or."""
        for key, cond in where.items():
            if key in ("$and", "$or"):
                if not all(isinstance(w, dict) and Collection._device_filter(w) for w in cond):
                    return False
            elif isinstance(cond, dict):
                if not all(op in _DEVICE_OPS for op in cond):
                    return False
            elif isinstance(cond, (list, dict)):
                return False
        return True

    def _where_mask(self, where):
        """uint8 [count] on the device.  Equality forms (what the reference's callers build,
        scripts/chromadb_helper.py:265-266) are evaluated there from cached per-key value codes; the rest
        ($gt.., $contains) by `_match` over the host metadata."""
        eng = engine()
        n = len(self.ids)
        if not self._device_filter(where):
            keep = np.fromiter((_match(m, where) for m in self.metadatas), np.uint8, n)
            return torch.from_numpy(keep).to(eng.device)
        mask = torch.empty(n, dtype=torch.uint8, device=eng.device)
        self._eval_where(where, mask, "set")
        return mask

    def _eval_where(self, where, mask, combine):
        """mask <- combine(mask, AND of the clauses of `where`)."""
        eng = engine()
        n = mask.numel()
        clauses = []  # (kind, payload): leaves first-class, groups through a temporary
        for key, cond in where.items():
            if key in ("$and", "$or"):
                clauses.append(("group", (key, cond)))
            elif isinstance(cond, dict):
                for op, arg in cond.items():
                    clauses.append(("leaf", (key, op, arg)))
            else:
                clauses.append(("leaf", (key, "$eq", cond)))
        target = mask
        if combine != "set" and len(clauses) != 1:
            target = torch.empty(n, dtype=torch.uint8, device=eng.device)  # the clauses AND into a temporary first
        if not clauses:
            target.fill_(1)
        for i, (kind, payload) in enumerate(clauses):
            comb = ("set" if target is not mask or combine == "set" else combine) if i == 0 else "and"
            if kind == "leaf":
                key, op, arg = payload
                st = self._key_codes(key)
                args = list(arg) if op in ("$in", "$nin") else [arg]
                vals = []
                for a in args:
                    try:
                        if a in st["map"]:
                            vals.append(st["map"][a])
                    except TypeError:
                        pass
                values = torch.tensor(vals, dtype=torch.int32).to(eng.device) if vals else None
                eng.row_mask_update(target, codes=st["dev"], values=values, negate=op in ("$ne", "$nin"), combine=comb)
            else:
                key, subs = payload
                tmp = torch.empty(n, dtype=torch.uint8, device=eng.device)
                if not subs:
                    tmp.fill_(1 if key == "$and" else 0)
                for j, w in enumerate(subs):
                    self._eval_where(w, tmp, "set" if j == 0 else ("and" if key == "$and" else "or"))
                eng.row_mask_update(target, other=tmp, combine=comb)
        if target is not mask:
            eng.row_mask_update(mask, other=target, combine=combine)

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
            corpus = self._resident()
            mask = self._where_mask(where) if where else None
            if k <= MAX_K:
                d, r = eng.score_topk(corpus, q.contiguous(), k, space=self.space, mask=mask)
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

    def query_keyword_classes(self, query_texts, keywords, n_results=10, include=None):
        """For the full-corpus hybrid mode (an extension, not reference behaviour): per query, the n_results nearest
        rows of EVERY keyword-match class -- class c holds the documents that contain exactly c of the query's
        `keywords[b]` as substrings of the lower-cased text, the test of scripts/chromadb_helper.py:352-359 -- merged
        nearest first.  Within a class the fused score of scripts/chromadb_helper.py:310 falls as the distance grows,
        so these (m + 1) * n_results rows contain the n_results best of the whole collection; the caller fuses them
        exactly as it fuses the 2n nearest.  Same result shape as query()."""
        include = list(include) if include is not None else ["documents", "metadatas", "distances"]
        eng = engine()
        texts = list(query_texts)
        q = self._ef.embed_tensor(texts) if hasattr(self._ef, "embed_tensor") else torch.tensor(self._ef(texts), dtype=torch.float32, device=eng.device)
        res = {"ids": [], "documents": [] if "documents" in include else None, "metadatas": [] if "metadatas" in include else None, "distances": [] if "distances" in include else None}
        n = self.count()
        k = min(int(n_results), n, MAX_K)
        if n == 0 or k <= 0:
            for _ in texts:
                for key in res:
                    if res[key] is not None:
                        res[key].append([])
            return res
        low = getattr(self, "_lower_docs", None)
        if low is None or len(low) != n:
            low = self._lower_docs = [(d or "").lower() for d in self.documents]
        pitch = (n + 3) // 4 * 4
        cls = np.zeros((len(texts), pitch), np.uint8)
        levels = 1
        most = max((len(list(kws)) for kws in keywords), default=0)
        if most > 255:
            raise ValueError(f"{most} keywords in one query: a row's match count is one byte (at most 255)")
        if (most + 1) * k > 2048:  # before any pass over the corpus
            raise ValueError(f"(keywords + 1) * n_results = {(most + 1) * k} candidates per query exceed 2048")
        for b, kws in enumerate(keywords):
            kws = list(kws)
            levels = max(levels, len(kws) + 1)
            if kws:
                cls[b, :n] = np.fromiter((sum(1 for kw in kws if kw in d) for d in low), np.uint8, n)
        corpus = self._resident()
        keys = eng.score_topk_levels(corpus, q.contiguous(), k, levels, torch.from_numpy(cls).to(eng.device), space=self.space)
        merged = eng.merge_topk(keys.view(1, len(texts), levels * k))  # all classes, nearest first
        dist, rows = eng.unpack_keys(merged)
        rows, dist = rows.cpu().numpy(), dist.cpu().numpy()
        for b in range(len(texts)):
            rr = [int(x) for x in rows[b] if x >= 0]
            res["ids"].append([self.ids[x] for x in rr])
            if res["documents"] is not None:
                res["documents"].append([self.documents[x] for x in rr])
            if res["metadatas"] is not None:
                res["metadatas"].append([self.metadatas[x] for x in rr])
            if res["distances"] is not None:
                res["distances"].append([float(dist[b][j]) for j in range(len(rr))])
        return res

    def keyword_stats(self, keywords):
        """Collection statistics for the BM25 keyword weighting (an extension): ({keyword: documents that contain it
        as a substring of their lower-cased text}, number of documents, mean document length in words).  One pass
        over the cached lower-cased documents per keyword not seen before."""
        n = self.count()
        low = getattr(self, "_lower_docs", None)
        if low is None or len(low) != n:
            low = self._lower_docs = [(d or "").lower() for d in self.documents]
        if getattr(self, "_kw_rows", None) != n:  # the statistics belong to this many documents
            self._avg_words = (sum(len(d.split()) for d in low) / n) if n else 0.0
            self._kw_df, self._kw_rows = {}, n
        for k in keywords:
            if k not in self._kw_df:
                self._kw_df[k] = sum(1 for d in low if k in d)
        return {k: self._kw_df[k] for k in keywords}, n, self._avg_words

    def _topk_beyond_max(self, q, k, mask):
        """More results than one scan returns (n_results is unbounded in the reference): passes of MAX_K for the
        whole batch, each resumed behind the last key every query has received (hrs_score_topk_after).  The order
        is total (distance, then row), so the passes concatenate to the exact top-k."""
        eng = engine()
        corpus = self._resident()
        d, r, keys = eng.score_topk(corpus, q, MAX_K, space=self.space, mask=mask, want_keys=True)
        dist, rows = [d], [r]
        got = MAX_K
        while got < k:
            step = min(MAX_K, k - got)
            d, r, keys = eng.score_topk_after(corpus, q, step, keys[:, -1].contiguous(), space=self.space, mask=mask)
            dist.append(d)
            rows.append(r)
            got += step
        return torch.cat(dist, dim=1), torch.cat(rows, dim=1)

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
        validate_name(name)
        if name not in self._open:
            self._open[name] = Collection(self, name, embedding_function, metadata)
        col = self._open[name]
        if embedding_function is not None:
            col._ef = embedding_function
        return col

    def list_collections(self):
        names = set(self._open)
        if self.path and os.path.isdir(self.path):
            names |= {d for d in os.listdir(self.path) if _NAME_RE.match(d) and os.path.exists(os.path.join(self.path, d, "meta.json"))}
        return [self.get_or_create_collection(n) for n in sorted(names)]
