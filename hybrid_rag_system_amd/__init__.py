"""MI355X-native retrieval engine behind the chromadb_helper embed / index / query surface.

Layout:
  csrc/        hand-written gfx950 HIP kernels + the C ABI (include/hrs.h) -> libhrs.so
  _native.py   ctypes binding (raises when the library is missing; there is no CPU path)
  engine.py    Engine / Corpus / Postings handles over torch-ROCm tensors
"""
from ._native import HrsError  # noqa: F401

__all__ = ["HrsError"]
