"""Host-side string work of the helper: cleaning, chunking, ids, metadata heuristics, keywords.

These stay on the CPU (micro-seconds per document) and reproduce the reference's behaviour exactly;
each function cites the reference lines it mirrors (scripts/chromadb_helper.py).
"""
import hashlib
import re

_WS = re.compile(r"\s+")
# break points, best first (chromadb_helper.py:142-148); the text is whitespace-collapsed before
# chunking, so the paragraph pattern can never fire -- it is kept because the reference tries it
_BREAK_RES = [re.compile(p) for p in (r"\n\n", r"\.\s+", r"[;:]\s+", r",\s+", r"\s+")]
_WORD = re.compile(r"\b[a-z]+\b")
_TITLE = re.compile(r"^#\s+(.+)$", re.MULTILINE)

STOPWORDS = frozenset((
    "the a an is are was were be been being have has had do does did will would could should may "
    "might can to of in for on with at by from as into through during before after above below "
    "between under again further then once here there when where why how all each few more most "
    "other some such no nor not only own same so than too very just and but if or because until "
    "while this that these those what which who whom my your his her its our their i me you he she "
    "it we they"
).split())

SYSTEM_RULES = {
    "unraid": (r"\bunraid\b", r"\b/mnt/user\b", r"\barray\b"),
    "proxmox": (r"\bproxmox\b", r"\bpve\b", r"\blxc\b", r"\bqemu\b"),
    "windows": (r"\bpowershell\b", r"\bactive directory\b", r"\bsccm\b"),
    "docker": (r"\bdocker\b", r"\bcontainer\b", r"\bcompose\b"),
    "network": (r"\bunifi\b", r"\bvlan\b", r"\bfirewall\b"),
}
_CATEGORY_RULES = (
    (("troubleshoot",), "troubleshooting"),
    (("howto", "guide"), "howto"),
    (("reference", "api"), "reference"),
    (("config",), "configuration"),
)


def clean_text(text):
    """chromadb_helper.py:93-99: re.sub(r"\\s+", " ", text), NULs dropped, stripped.  str.split() splits on
    the same characters as \\s (str.isspace) and is several times faster than the regex; the one space
    the regex leaves at either end goes with the final strip."""
    return " ".join(text.split()).replace("\x00", "").strip()


def chunk_text(text, chunk_size, overlap):
    """chromadb_helper.py:102-173: windows of chunk_size chars, cut at the last break point found in
    the final fifth of the window, next window starting `overlap` chars before the cut."""
    text = clean_text(text)
    total = len(text)
    if total <= chunk_size:
        return [{"content": text, "chunk_index": 0, "char_start": 0, "char_end": total, "total_chunks": 1}]
    tail = int(chunk_size * 0.2)
    pieces = []
    pos = 0
    while pos < total:
        cut = pos + chunk_size
        if cut < total:
            base = cut - tail
            view = text[base:cut]
            for rx in _BREAK_RES:
                ends = [m.end() for m in rx.finditer(view)]
                if ends:
                    cut = base + ends[-1]
                    break
        body = text[pos:cut].strip()
        if body:
            pieces.append({"content": body, "chunk_index": len(pieces), "char_start": pos, "char_end": cut})
        pos = cut - overlap
    for p in pieces:
        p["total_chunks"] = len(pieces)
    return pieces


_SECTION_HEADING = re.compile(r'(^##?\s+.+$)', re.MULTILINE)


def chunk_document(content, chunk_size=2000, overlap=200):
    """The multi-system indexer's chunker (reference scripts/index-all-systems.py:355-387; defaults from its
    RAG_CHUNK_SIZE / RAG_CHUNK_OVERLAP, :39-40).  The text is cut before every level-1/2 markdown heading; pieces are
    packed greedily into chunks of at most chunk_size characters (a piece that does not fit starts a new chunk, which
    begins with the last `overlap` characters of the one just closed); chunks are stripped.  If that yields nothing,
    or a chunk longer than 1.5 x chunk_size (a section without headings), the text is cut into plain windows of
    chunk_size characters every chunk_size - overlap instead (blank windows dropped).  Never returns an empty list."""
    pieces = _SECTION_HEADING.split(content)
    chunks, cur = [], ""
    for piece in pieces:
        if cur and len(cur) + len(piece) > chunk_size:
            chunks.append(cur.strip())
            cur = (cur[-overlap:] if overlap > 0 else "") + piece
        else:
            cur += piece
    if cur.strip():
        chunks.append(cur.strip())
    if not chunks or max(len(c) for c in chunks) > chunk_size * 1.5:
        step = chunk_size - overlap
        chunks = [w.strip() for w in (content[i:i + chunk_size] for i in range(0, len(content), step)) if w.strip()]
    return chunks if chunks else [content]


def generate_doc_id(content, source=""):
    """chromadb_helper.py:223-226."""
    return hashlib.sha256((source + ":" + content[:500]).encode()).hexdigest()[:16]


def extract_metadata_from_content(content, filepath=None):
    """chromadb_helper.py:176-220."""
    meta = {}
    lowered = content.lower()
    # every pattern is \bliteral\b: the literal must occur for the regex to match, and `in` is much faster
    systems = [name for name, pats in SYSTEM_RULES.items() if any(p[2:-2] in lowered and re.search(p, lowered) for p in pats)]
    if systems:
        meta["system"] = systems[0]
        meta["systems"] = ",".join(systems)
    if filepath:
        low = filepath.lower()
        meta["category"] = next((cat for keys, cat in _CATEGORY_RULES if any(k in low for k in keys)), "general")
        meta["filepath"] = filepath
    m = _TITLE.search(content)
    if m:
        meta["title"] = m.group(1).strip()[:200]
    return meta


def extract_keywords(text):
    """chromadb_helper.py:328-349 (a set turned into a list: order is not part of the contract)."""
    return list({w for w in _WORD.findall(text.lower()) if len(w) > 2 and w not in STOPWORDS})


def calculate_keyword_score(document, keywords):
    """chromadb_helper.py:352-359: fraction of keywords that occur as substrings of the document."""
    if not keywords:
        return 0.0
    hay = document.lower()
    return sum(1 for k in keywords if k in hay) / len(keywords)


def bm25_keyword_score(document, keywords, doc_freq, n_docs, avg_len, k1=1.2, b=0.75):
    """An EXTENSION (the reference has only the fraction above): the keywords weighed BM25-style and normalised to
    [0, 1) so that it can take the fraction's place in the fused score.  tf = occurrences of the keyword as a substring
    of the lower-cased document (the containment test of chromadb_helper.py:352-359, counted), document length in
    whitespace-separated words, idf = ln(1 + (N - df + 0.5) / (df + 0.5)) with df = documents containing the keyword:
      sum idf * tf * (k1 + 1) / (tf + k1 * (1 - b + b * len / avg_len))  /  sum idf * (k1 + 1)."""
    import math

    if not keywords:
        return 0.0
    hay = document.lower()
    norm = k1 * ((1.0 - b) + b * (len(hay.split()) / avg_len if avg_len > 0 else 1.0))
    score = best = 0.0
    for k in keywords:
        w = math.log(1.0 + (n_docs - doc_freq.get(k, 0) + 0.5) / (doc_freq.get(k, 0) + 0.5))
        best += w * (k1 + 1.0)
        f = hay.count(k)
        if f > 0:
            score += w * f * (k1 + 1.0) / (f + norm)
    return score / best if best > 0.0 else 0.0
