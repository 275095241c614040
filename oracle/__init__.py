"""CPU oracle for the retrieval hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a checker: a plain restatement, on the CPU, of
what the reference's embed / index / query path computes.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The shipped package (``hybrid_rag_system_amd``) never imports
anything from here and has no CPU fallback: without the HIP library it raises.

Pinning status
--------------
* ``text_oracle``  -- pinned against the reference's own functions
  (``scripts/chromadb_helper.py``, ``scripts/mcp_chromadb_bridge.py``) imported
  in the build container; the vectors live in ``tests/golden/*.json`` and were
  written by ``tests/golden/make_goldens.py``.
* ``search_oracle`` (C, ``search_oracle.c``) -- the arithmetic below the
  ``collection.query`` seam lives in ChromaDB (``chromadb>=0.4.0``, unpinned in
  the reference's requirements.txt:3), which is not vendored in the reference
  and not installed here: **parity unpinned** against ChromaDB itself.  The
  semantics are the ones this build defines (exact fp64 distance from fp32
  inputs, rounded once to fp32; order = distance ascending, then row
  ascending); the reference-side consumers of those results (fusion, sorting,
  formatting) are pinned.
* ``encoder_oracle`` -- sentence-transformers / MiniLM weights are absent and
  cannot be fetched: **parity unpinned** against the real model.  The oracle is
  an fp32 restatement of the BERT/MiniLM-L6 forward with generator-seeded
  weights, checked against a config-constructed ``transformers.BertModel``
  holding the same weights, live and through the committed vectors that model
  produced (``tests/golden/encoder_goldens.npz``).
"""
