#!/usr/bin/env python3
"""Generate tests/golden/chunk_document_goldens.json by running the REFERENCE's own chunk_document
(scripts/index-all-systems.py:355-387).  That script imports chromadb at module level, which does not exist here, so
the one function is taken out of the file's syntax tree at generation time and executed with the module's CONFIG
defaults (chunk_size 2000, overlap 200, :39-40) -- nothing of the reference's text is kept: the output holds input
descriptions (how each input is built from the fixture files or a seed) and the expected chunks' lengths and hashes.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_chunk_document_goldens.py
"""
import ast
import hashlib
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.chunk_document_cases import build_cases  # noqa: E402

REF = os.environ.get("HRS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

src = open(os.path.join(REF, "scripts", "index-all-systems.py")).read()
fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "chunk_document")
ns = {"re": re, "CONFIG": {"chunk_size": 2000, "chunk_overlap": 200}}
exec(compile(ast.Module(body=[fn], type_ignores=[]), "index-all-systems.py", "exec"), ns)
ref_chunk = ns["chunk_document"]

out = {}
for name, (text, kw) in build_cases().items():
    chunks = ref_chunk(text, **kw)
    out[name] = {"n": len(chunks), "lens": [len(c) for c in chunks],
                 "sha": [hashlib.sha256(c.encode()).hexdigest()[:16] for c in chunks]}
with open(os.path.join(HERE, "chunk_document_goldens.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(len(out), "cases")
