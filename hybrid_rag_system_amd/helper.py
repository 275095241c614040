#!/usr/bin/env python3
"""Drop-in for the reference's scripts/chromadb_helper.py: same functions, same CLI, same JSON,
with the embedding model and the vector store replaced by the MI355X engine in this package.

  python -m hybrid_rag_system_amd.helper query "text" [-n N] [-c COLL] [--system S] [--hybrid] [--json]
  python -m hybrid_rag_system_amd.helper add "content" [-s SOURCE] [-c COLL]
  python -m hybrid_rag_system_amd.helper index PATH [-c COLL] [-r] [-p PATTERN ...]
  python -m hybrid_rag_system_amd.helper list-collections | stats [-c COLL]

Configuration keeps the reference's environment names (CHROMADB_COLLECTION, EMBEDDING_MODEL,
CHUNK_SIZE, CHUNK_OVERLAP; CHROMADB_HOST / CHROMADB_PORT are accepted and unused: there is no
server).  HRS_DATA_DIR names the directory that persists collections between invocations,
HRS_MODEL_DIR a local model directory, HRS_DEVICE the GPU.  With HRS_SERVER set to the Unix socket of a
resident engine (`python -m hybrid_rag_system_amd.server`), collections are reached through it, as the
reference reaches its Chroma server, and this process loads neither torch nor the library.  Additions
beside the reference surface: `embed`, `query_batch`, `hybrid_query_batch`.
"""
import argparse
import json
import os
from datetime import datetime, timezone
from pathlib import Path

from . import textproc

# ---- configuration (names and defaults of scripts/chromadb_helper.py:33-40) -----------------------
CHROMADB_HOST = os.getenv("CHROMADB_HOST", "<ADD-IP-ADDRESS>")
CHROMADB_PORT = int(os.getenv("CHROMADB_PORT", "8000"))
DEFAULT_COLLECTION = os.getenv("CHROMADB_COLLECTION", "infrastructure_docs")
EMBEDDING_MODEL = os.getenv("EMBEDDING_MODEL", "all-MiniLM-L6-v2")
CHUNK_SIZE = int(os.getenv("CHUNK_SIZE", "1000"))
CHUNK_OVERLAP = int(os.getenv("CHUNK_OVERLAP", "200"))
DATA_DIR = os.getenv("HRS_DATA_DIR", os.path.join(os.path.expanduser("~"), ".hrs", "collections"))

_embedding_function = None
_client = None

clean_text = textproc.clean_text
extract_metadata_from_content = textproc.extract_metadata_from_content
generate_doc_id = textproc.generate_doc_id
extract_keywords = textproc.extract_keywords
calculate_keyword_score = textproc.calculate_keyword_score
bm25_keyword_score = textproc.bm25_keyword_score  # (an extension: see hybrid_query)


def chunk_text(text, chunk_size=None, overlap=None):
    return textproc.chunk_text(text, CHUNK_SIZE if chunk_size is None else chunk_size, CHUNK_OVERLAP if overlap is None else overlap)


def get_embedding_function():
    """Process-wide encoder, created on first use (scripts/chromadb_helper.py:58-71)."""
    global _embedding_function
    if _embedding_function is None:
        from .store import HrsEmbeddingFunction

        _embedding_function = HrsEmbeddingFunction(model_name=EMBEDDING_MODEL)
    return _embedding_function


def _server_address():
    return os.getenv("HRS_SERVER") or None


def get_client():
    """scripts/chromadb_helper.py:74-76: the in-process store, or the resident engine named by HRS_SERVER
    (an unreachable server raises, as an unreachable Chroma server does in the reference)."""
    global _client
    if _client is None:
        if _server_address():
            from .remote import RemoteClient

            _client = RemoteClient(_server_address())
        else:
            from .store import Client

            _client = Client(DATA_DIR)
    return _client


def get_collection(name=None):
    """scripts/chromadb_helper.py:79-86."""
    return get_client().get_or_create_collection(
        name=name or DEFAULT_COLLECTION,
        embedding_function=None if _server_address() else get_embedding_function(),  # the server embeds
        metadata={"description": "Infrastructure and homelab documentation"},
    )


# ---- queries -------------------------------------------------------------------------------------
def query(query_text, n_results=5, collection_name=None, where_filter=None, include_distances=True):
    """Semantic search (scripts/chromadb_helper.py:233-272): the collection's raw result dict, or {"error": ...}."""
    collection = get_collection(collection_name)
    params = {"query_texts": [query_text], "n_results": n_results, "include": ["documents", "metadatas"] + (["distances"] if include_distances else [])}
    if where_filter:
        params["where"] = where_filter
    try:
        return collection.query(**params)
    except Exception as e:  # the reference swallows everything here
        return {"error": str(e)}


def _fuse(documents, metadatas, ids, distances, keywords, n_results, keyword_boost, keyword_score=calculate_keyword_score):
    """The scoring loop, sort and cut of scripts/chromadb_helper.py:299-325 for one query."""
    scored = []
    for i, doc in enumerate(documents):
        distance = distances[i] if distances is not None else 1.0
        semantic = max(0, 1 - distance)
        kw = keyword_score(doc, keywords)
        scored.append({
            "document": doc,
            "metadata": metadatas[i] if metadatas is not None else {},
            "id": ids[i] if ids is not None else None,
            "semantic_score": semantic,
            "keyword_score": kw,
            "combined_score": semantic * (1 - keyword_boost) + kw * keyword_boost,
            "distance": distance,
        })
    scored.sort(key=lambda r: r["combined_score"], reverse=True)
    return scored[:n_results]


def hybrid_query(query_text, n_results=5, collection_name=None, keyword_boost=0.3, full_corpus=None, keyword_scoring=None):
    """Semantic top-2n re-ranked with the keyword fraction (scripts/chromadb_helper.py:275-325).
    Like the reference it does not catch exceptions and ignores any `where` filter.

    full_corpus (default: the environment's HRS_HYBRID_FULL == "1", i.e. off) switches to an EXTENSION the reference
    does not have: the same fused score, but its n best over the whole collection instead of over the 2n nearest
    rows -- a document that contains every keyword is found even when it is not among the 2n nearest.

    keyword_scoring (default: the environment's HRS_KEYWORD_SCORE, else "fraction") = "bm25" is a second EXTENSION:
    the candidates' keyword score is textproc.bm25_keyword_score (term frequency, document length and rarity of a
    keyword in the collection) instead of the reference's fraction of keywords present.  It applies to the re-rank of
    the 2n nearest; with full_corpus the classes are still the reference's match counts."""
    collection = get_collection(collection_name)
    keywords = extract_keywords(query_text)
    keyword_score = _keyword_score_fn(collection, keywords, keyword_scoring)
    if full_corpus is None:
        full_corpus = os.getenv("HRS_HYBRID_FULL") == "1"
    if full_corpus and keywords:
        res = collection.query_keyword_classes([query_text], [keywords], n_results, include=["documents", "metadatas", "distances"])
    else:
        res = collection.query(query_texts=[query_text], n_results=n_results * 2, include=["documents", "metadatas", "distances"])
    if not (res.get("documents") and res["documents"][0]):
        return []
    return _fuse(
        res["documents"][0],
        res["metadatas"][0] if res.get("metadatas") else None,
        res["ids"][0] if res.get("ids") else None,
        res["distances"][0] if res.get("distances") else None,
        keywords,
        n_results,
        keyword_boost,
        keyword_score,
    )


# ---- batched additions (not in the reference: it has one query per process) -------------------------
def embed(texts):
    """List[str] -> fp32 [n, 384] CUDA tensor of unit rows (through a resident engine: nested lists)."""
    if _server_address():
        return get_client().embed(texts)
    return get_embedding_function().embed_tensor(list(texts))


def query_batch(query_texts, n_results=5, collection_name=None, where_filter=None):
    collection = get_collection(collection_name)
    params = {"query_texts": list(query_texts), "n_results": n_results, "include": ["documents", "metadatas", "distances"]}
    if where_filter:
        params["where"] = where_filter
    return collection.query(**params)


def _keyword_score_fn(collection, keywords, keyword_scoring):
    """The candidates' keyword score: the reference's fraction, or (an extension) BM25 over the collection's statistics."""
    scoring = keyword_scoring or os.getenv("HRS_KEYWORD_SCORE") or "fraction"
    if scoring not in ("fraction", "bm25"):
        raise ValueError(f"keyword_scoring {scoring!r}: 'fraction' or 'bm25'")
    if scoring == "bm25" and keywords:
        df, n_docs, avg_len = collection.keyword_stats(keywords)
        return lambda doc, kws: bm25_keyword_score(doc, kws, df, n_docs, avg_len)
    return calculate_keyword_score


def hybrid_query_batch(query_texts, n_results=5, collection_name=None, keyword_boost=0.3, keyword_scoring=None):
    query_texts = list(query_texts)
    res = query_batch(query_texts, n_results * 2, collection_name)
    collection = get_collection(collection_name)
    out = []
    for b, q in enumerate(query_texts):
        kws = extract_keywords(q)
        out.append(_fuse(res["documents"][b], res["metadatas"][b], res["ids"][b], res["distances"][b], kws, n_results, keyword_boost,
                         _keyword_score_fn(collection, kws, keyword_scoring)))
    return out


# ---- add / index -----------------------------------------------------------------------------------
def _now():
    return datetime.now(timezone.utc).isoformat()


def add_document(content, metadata=None, doc_id=None, collection_name=None):
    """scripts/chromadb_helper.py:366-404."""
    collection = get_collection(collection_name)
    if doc_id is None:
        doc_id = generate_doc_id(content, metadata.get("source", "") if metadata else "")
    full = extract_metadata_from_content(content)
    if metadata:
        full.update(metadata)
    full["indexed_at"] = _now()
    try:
        collection.add(documents=[content], metadatas=[full], ids=[doc_id])
        return {"status": "success", "id": doc_id}
    except Exception as e:
        return {"status": "error", "error": str(e)}


def _prepare_file(filepath, chunk=True):
    """Read and chunk one file -> (path, docs, metas, ids, n_chunks) or an error dict (scripts/chromadb_helper.py:407-459)."""
    path = Path(filepath)
    if not path.exists():
        return {"status": "error", "error": f"File not found: {filepath}"}
    try:
        content = path.read_text(encoding="utf-8", errors="ignore")
    except Exception as e:
        return {"status": "error", "error": f"Failed to read file: {e}"}
    base = {"source": path.name, "filepath": str(path.absolute()), "file_type": path.suffix.lstrip("."), "indexed_at": _now()}
    base.update(extract_metadata_from_content(content, str(path)))
    if chunk and len(content) > CHUNK_SIZE:
        chunks = chunk_text(content)
        stem = generate_doc_id(content, str(path))
        ids = [f"{stem}-{i:04d}" for i in range(len(chunks))]
        metas = [dict(base, chunk_index=c["chunk_index"], total_chunks=c["total_chunks"]) for c in chunks]
        docs = [c["content"] for c in chunks]
    else:
        chunks = [None]
        ids, metas, docs = [generate_doc_id(content, str(path))], [base], [content]
    return (path, docs, metas, ids, len(chunks))


def index_file(filepath, collection_name=None, chunk=True, _prepared=None, _embeddings=None):
    """scripts/chromadb_helper.py:407-481: one collection.add per file."""
    prep = _prepared if _prepared is not None else _prepare_file(filepath, chunk)
    if isinstance(prep, dict):
        return prep
    path, docs, metas, ids, n_chunks = prep
    collection = get_collection(collection_name)
    try:
        if _embeddings is None:
            collection.add(documents=docs, metadatas=metas, ids=ids)  # the reference's call, verbatim
        else:
            collection.add(documents=docs, metadatas=metas, ids=ids, embeddings=_embeddings)
        return {"status": "success", "chunks": n_chunks, "filepath": str(path)}
    except Exception as e:
        return {"status": "error", "error": str(e)}


def index_directory(directory, patterns=None, collection_name=None, recursive=True):
    """scripts/chromadb_helper.py:484-529.  Same files, order, ids and per-file results as indexing them
    one by one; the chunks of all files are embedded together (full encoder batches) before the per-file
    collection.add calls, which then receive their embeddings."""
    if patterns is None:
        patterns = ["*.md", "*.txt", "*.yaml", "*.yml", "*.json"]
    root = Path(directory)
    if not root.exists():
        return {"status": "error", "error": f"Directory not found: {directory}"}
    results = {"status": "success", "indexed": 0, "failed": 0, "files": []}
    files = [fp for pattern in patterns for fp in (root.rglob(pattern) if recursive else root.glob(pattern))]
    prepared = [_prepare_file(str(fp)) for fp in files]
    embeddings = None
    try:
        ef = get_collection(collection_name)._ef
        if hasattr(ef, "embed_tensor"):
            texts = [d for p in prepared if not isinstance(p, dict) for d in p[1]]
            embeddings = ef.embed_tensor(texts).cpu().numpy() if texts else None
    except Exception:
        embeddings = None  # the per-file adds embed for themselves and report their own errors
    at = 0
    for filepath, prep in zip(files, prepared):
        emb = None
        if not isinstance(prep, dict) and embeddings is not None:
            emb = embeddings[at : at + len(prep[1])]
            at += len(prep[1])
        r = index_file(str(filepath), collection_name, _prepared=prep, _embeddings=emb)
        if r["status"] == "success":
            results["indexed"] += 1
            results["files"].append({"path": str(filepath), "chunks": r.get("chunks", 1)})
        else:
            results["failed"] += 1
            print(f"  Failed: {filepath}: {r.get('error', 'unknown')}")
    return results


# ---- collection management ---------------------------------------------------------------------------
def list_collections():
    """scripts/chromadb_helper.py:536-554."""
    out = []
    for col in get_client().list_collections():
        try:
            count = col.count()
        except Exception:
            count = "unknown"
        out.append({"name": col.name, "count": count, "metadata": col.metadata})
    return out


def get_collection_stats(collection_name=None):
    """scripts/chromadb_helper.py:557-584."""
    collection = get_collection(collection_name)
    try:
        count = collection.count()
        sample = collection.peek(limit=100)
        systems, categories = {}, {}
        for meta in sample.get("metadatas", []):
            if meta:
                s, c = meta.get("system", "unknown"), meta.get("category", "unknown")
                systems[s] = systems.get(s, 0) + 1
                categories[c] = categories.get(c, 0) + 1
        return {"name": collection.name, "count": count, "systems_sample": systems, "categories_sample": categories}
    except Exception as e:
        return {"error": str(e)}


# ---- CLI (scripts/chromadb_helper.py:591-684) --------------------------------------------------------
def _parser():
    parser = argparse.ArgumentParser(description="ChromaDB Helper")
    sub = parser.add_subparsers(dest="command", help="Command")
    q = sub.add_parser("query", help="Search the knowledge base")
    q.add_argument("text", help="Search query")
    q.add_argument("-n", "--n-results", type=int, default=5)
    q.add_argument("-c", "--collection", default=DEFAULT_COLLECTION)
    q.add_argument("--system", help="Filter by system (unraid, proxmox, windows, etc)")
    q.add_argument("--hybrid", action="store_true", help="Use hybrid search")
    q.add_argument("--json", action="store_true", help="Output as JSON")
    a = sub.add_parser("add", help="Add a document")
    a.add_argument("content", help="Document content")
    a.add_argument("-s", "--source", help="Source identifier")
    a.add_argument("-c", "--collection", default=DEFAULT_COLLECTION)
    i = sub.add_parser("index", help="Index a file or directory")
    i.add_argument("path", help="File or directory path")
    i.add_argument("-c", "--collection", default=DEFAULT_COLLECTION)
    i.add_argument("-r", "--recursive", action="store_true", default=True)
    i.add_argument("-p", "--patterns", nargs="+", help="File patterns to match")
    sub.add_parser("list-collections", help="List all collections")
    s = sub.add_parser("stats", help="Get collection statistics")
    s.add_argument("-c", "--collection", default=DEFAULT_COLLECTION)
    return parser


def _print_hit(header, meta, doc):
    print(header)
    print(f"Source: {meta.get('source', 'unknown')}")
    print(f"System: {meta.get('system', 'unknown')}")
    print(f"{doc[:500]}...")
    print()


def main(argv=None):
    parser = _parser()
    args = parser.parse_args(argv)
    if args.command == "query":
        where = {"system": args.system} if args.system else None
        if args.hybrid:
            results = hybrid_query(args.text, args.n_results, args.collection)
            if args.json:
                print(json.dumps(results, indent=2))
            else:
                print(f"\nHybrid search results for: {args.text}\n")
                for n, r in enumerate(results):
                    _print_hit(f"--- Result {n+1} (score: {r['combined_score']:.3f}) ---", r["metadata"], r["document"])
        else:
            results = query(args.text, args.n_results, args.collection, where)
            if args.json:
                print(json.dumps(results, indent=2))
            else:
                print(f"\nSearch results for: {args.text}\n")
                if results.get("documents") and results["documents"][0]:
                    for n, doc in enumerate(results["documents"][0]):
                        meta = results["metadatas"][0][n] if results.get("metadatas") else {}
                        dist = results["distances"][0][n] if results.get("distances") else "N/A"
                        # the reference's format spec on this line is malformed and raises at run time
                        # (scripts/chromadb_helper.py:653); the evident intent is printed instead
                        shown = f"{dist:.3f}" if isinstance(dist, float) else dist
                        _print_hit(f"--- Result {n+1} (distance: {shown}) ---", meta, doc)
                else:
                    print("No results found")
    elif args.command == "add":
        print(json.dumps(add_document(args.content, {"source": args.source} if args.source else {}, collection_name=args.collection), indent=2))
    elif args.command == "index":
        path = Path(args.path)
        if path.is_file():
            result = index_file(str(path), args.collection)
        else:
            result = index_directory(str(path), args.patterns, args.collection, args.recursive)
        print(json.dumps(result, indent=2))
    elif args.command == "list-collections":
        for col in list_collections():
            print(f"  - {col['name']} ({col['count']} documents)")
    elif args.command == "stats":
        print(json.dumps(get_collection_stats(args.collection), indent=2))
    else:
        parser.print_help()


if __name__ == "__main__":
    main()
