"""Text-side oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Plain-Python restatement of the host-side string work on the reference's
embed / index / query path.  Each function names the reference lines it
follows (paths relative to the reference checkout).  Pinned by
tests/golden/text_goldens.json, which was produced by the reference itself.
"""
import hashlib
import re

# scripts/chromadb_helper.py:331-343 -- the literal stop-word set (101 words).
_STOP = frozenset(
    "the a an is are was were be been being have has had do does did will would could "
    "should may might can to of in for on with at by from as into through during before after "
    "above below between under again further then once here there when where why how all each few "
    "more most other some such no nor not only own same so than too very just and but if "
    "or because until while this that these those what which who whom my your his her its "
    "our their i me you he she it we they".split()
)

# scripts/chromadb_helper.py:142-148 -- break patterns tried in this order.
_BREAKS = (r"\n\n", r"\.\s+", r"[;:]\s+", r",\s+", r"\s+")

# scripts/chromadb_helper.py:181-187 -- system detection rules (dict order matters).
_SYSTEMS = (
    ("unraid", (r"\bunraid\b", r"\b/mnt/user\b", r"\barray\b")),
    ("proxmox", (r"\bproxmox\b", r"\bpve\b", r"\blxc\b", r"\bqemu\b")),
    ("windows", (r"\bpowershell\b", r"\bactive directory\b", r"\bsccm\b")),
    ("docker", (r"\bdocker\b", r"\bcontainer\b", r"\bcompose\b")),
    ("network", (r"\bunifi\b", r"\bvlan\b", r"\bfirewall\b")),
)


def clean(text):
    """scripts/chromadb_helper.py:93-99."""
    return re.sub(r"\s+", " ", text).replace("\x00", "").strip()


def chunk(text, size=1000, overlap=200):
    """scripts/chromadb_helper.py:102-173."""
    text = clean(text)
    n = len(text)
    if n <= size:
        return [dict(content=text, chunk_index=0, char_start=0, char_end=n, total_chunks=1)]
    out, start = [], 0
    while start < n:
        end = start + size
        if end < n:
            lo = end - int(size * 0.2)
            window = text[lo:end]
            for pat in _BREAKS:
                last = None
                for last in re.finditer(pat, window):
                    pass
                if last is not None:
                    end = lo + last.end()
                    break
        piece = text[start:end].strip()
        if piece:
            out.append(dict(content=piece, chunk_index=len(out), char_start=start, char_end=end))
        start = end - overlap
    for c in out:
        c["total_chunks"] = len(out)
    return out


def doc_id(content, source=""):
    """scripts/chromadb_helper.py:223-226."""
    return hashlib.sha256(f"{source}:{content[:500]}".encode()).hexdigest()[:16]


def content_metadata(content, filepath=None):
    """scripts/chromadb_helper.py:176-220."""
    md = {}
    low = content.lower()
    hits = [name for name, pats in _SYSTEMS if any(re.search(p, low) for p in pats)]
    if hits:
        md["system"] = hits[0]
        md["systems"] = ",".join(hits)
    if filepath:
        p = filepath.lower()
        if "troubleshoot" in p:
            cat = "troubleshooting"
        elif "howto" in p or "guide" in p:
            cat = "howto"
        elif "reference" in p or "api" in p:
            cat = "reference"
        elif "config" in p:
            cat = "configuration"
        else:
            cat = "general"
        md["category"] = cat
        md["filepath"] = filepath
    m = re.search(r"^#\s+(.+)$", content, re.MULTILINE)
    if m:
        md["title"] = m.group(1).strip()[:200]
    return md


def keywords(text):
    """scripts/chromadb_helper.py:328-349 (returned sorted: the reference's order is hash-seed dependent)."""
    return sorted({w for w in re.findall(r"\b[a-z]+\b", text.lower()) if w not in _STOP and len(w) > 2})


def keyword_score(document, kws):
    """scripts/chromadb_helper.py:352-359 -- substring presence count / number of keywords."""
    if not kws:
        return 0.0
    low = document.lower()
    return sum(1 for k in kws if k in low) / len(kws)


def fuse(documents, metadatas, ids, distances, query_text, n_results, boost=0.3):
    """scripts/chromadb_helper.py:296-325 -- score the top-2n candidates, stable-sort, cut to n."""
    kws = keywords(query_text)
    rows = []
    for i, doc in enumerate(documents):
        d = distances[i] if distances is not None else 1.0
        sem = max(0, 1 - d)
        kw = keyword_score(doc, kws)
        rows.append(
            dict(
                document=doc,
                metadata=metadatas[i] if metadatas is not None else {},
                id=ids[i] if ids is not None else None,
                semantic_score=sem,
                keyword_score=kw,
                combined_score=sem * (1 - boost) + kw * boost,
                distance=d,
            )
        )
    rows.sort(key=lambda r: r["combined_score"], reverse=True)
    return rows[:n_results]


def bridge_plain(results, max_len=800):
    """scripts/mcp_chromadb_bridge.py:33-64."""
    if "error" in results:
        return {"success": False, "error": results["error"]}
    out = {"success": True, "count": 0, "results": []}
    docs = results.get("documents")
    if docs and docs[0]:
        out["count"] = len(docs[0])
        for i, doc in enumerate(docs[0]):
            item = {"index": i + 1, "document": doc[:max_len] + ("..." if len(doc) > max_len else ""), "full_length": len(doc)}
            if results.get("metadatas") and results["metadatas"][0]:
                item["metadata"] = results["metadatas"][0][i]
            if results.get("distances") and results["distances"][0]:
                d = results["distances"][0][i]
                item["distance"] = round(d, 4)
                item["similarity"] = round(max(0, 1 - d), 4)
            out["results"].append(item)
    return out


def bridge_hybrid(rows, max_len=800):
    """scripts/mcp_chromadb_bridge.py:67-88."""
    out = {"success": True, "count": len(rows), "results": []}
    for i, r in enumerate(rows):
        doc = r["document"]
        out["results"].append(
            {
                "index": i + 1,
                "document": doc[:max_len] + ("..." if len(doc) > max_len else ""),
                "full_length": len(doc),
                "metadata": r.get("metadata", {}),
                "combined_score": round(r.get("combined_score", 0), 4),
                "semantic_score": round(r.get("semantic_score", 0), 4),
                "keyword_score": round(r.get("keyword_score", 0), 4),
            }
        )
    return out
