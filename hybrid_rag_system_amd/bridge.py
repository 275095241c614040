#!/usr/bin/env python3
"""Drop-in for the reference's scripts/mcp_chromadb_bridge.py: the JSON contract the MCP server's
`chromadb_query` tool consumes (mcp-server/src/index.ts:159-171, docs/LOCAL-LLM-IMPROVEMENTS.md:76-107).

  python -m hybrid_rag_system_amd.bridge query "text" [--hybrid] [--n-results N] [--collection C] [--system S] [--max-length L]
  python -m hybrid_rag_system_amd.bridge add "content" [--source S] [--collection C] [--system S] [--category C]
  python -m hybrid_rag_system_amd.bridge collections | stats [--collection C]

One compact JSON object per invocation on stdout; any exception becomes {"success": false, "error": ...}
with exit status 1 (scripts/mcp_chromadb_bridge.py:156-158).
"""
import argparse
import json
import sys

from . import helper


def _clip(doc, limit):
    return doc[:limit] + ("..." if len(doc) > limit else "")


def format_query_results(results, max_doc_length=800):
    """scripts/mcp_chromadb_bridge.py:33-64."""
    if "error" in results:
        return {"success": False, "error": results["error"]}
    out = {"success": True, "count": 0, "results": []}
    docs = results.get("documents")
    if docs and docs[0]:
        out["count"] = len(docs[0])
        metas = results.get("metadatas")
        dists = results.get("distances")
        for n, doc in enumerate(docs[0]):
            item = {"index": n + 1, "document": _clip(doc, max_doc_length), "full_length": len(doc)}
            if metas and metas[0]:
                item["metadata"] = metas[0][n]
            if dists and dists[0]:
                item["distance"] = round(dists[0][n], 4)
                item["similarity"] = round(max(0, 1 - dists[0][n]), 4)
            out["results"].append(item)
    return out


def format_hybrid_results(results, max_doc_length=800):
    """scripts/mcp_chromadb_bridge.py:67-88."""
    return {
        "success": True,
        "count": len(results),
        "results": [
            {
                "index": n + 1,
                "document": _clip(r["document"], max_doc_length),
                "full_length": len(r["document"]),
                "metadata": r.get("metadata", {}),
                "combined_score": round(r.get("combined_score", 0), 4),
                "semantic_score": round(r.get("semantic_score", 0), 4),
                "keyword_score": round(r.get("keyword_score", 0), 4),
            }
            for n, r in enumerate(results)
        ],
    }


def _parser():
    parser = argparse.ArgumentParser(description="MCP ChromaDB Bridge")
    sub = parser.add_subparsers(dest="command")
    q = sub.add_parser("query")
    q.add_argument("text", help="Search query")
    q.add_argument("--n-results", type=int, default=5)
    q.add_argument("--collection", default="infrastructure_docs")
    q.add_argument("--system", help="Filter by system")
    q.add_argument("--hybrid", action="store_true")
    q.add_argument("--max-length", type=int, default=800)
    a = sub.add_parser("add")
    a.add_argument("content", help="Document content")
    a.add_argument("--source", default="mcp")
    a.add_argument("--collection", default="infrastructure_docs")
    a.add_argument("--system", help="System tag")
    a.add_argument("--category", help="Category tag")
    sub.add_parser("collections")
    s = sub.add_parser("stats")
    s.add_argument("--collection", default="infrastructure_docs")
    return parser


def main(argv=None):
    args = _parser().parse_args(argv)
    try:
        if args.command == "query":
            if args.hybrid:
                output = format_hybrid_results(helper.hybrid_query(args.text, args.n_results, args.collection), args.max_length)
            else:
                where = {"system": args.system} if args.system else None
                output = format_query_results(helper.query(args.text, args.n_results, args.collection, where), args.max_length)
            print(json.dumps(output))
        elif args.command == "add":
            metadata = {"source": args.source}
            if args.system:
                metadata["system"] = args.system
            if args.category:
                metadata["category"] = args.category
            print(json.dumps(helper.add_document(args.content, metadata, collection_name=args.collection)))
        elif args.command == "collections":
            print(json.dumps({"success": True, "collections": helper.list_collections()}))
        elif args.command == "stats":
            print(json.dumps({"success": True, "stats": helper.get_collection_stats(args.collection)}))
        else:
            print(json.dumps({"success": False, "error": "Unknown command"}))
            sys.exit(1)
    except Exception as e:
        print(json.dumps({"success": False, "error": str(e)}))
        sys.exit(1)


if __name__ == "__main__":
    main()
