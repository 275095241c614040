#!/usr/bin/env python3
"""`import chromadb_helper` / `python chromadb_helper.py ...` keep working: everything lives in
hybrid_rag_system_amd.helper (the MI355X-backed mirror of the reference's module of this name)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from hybrid_rag_system_amd.helper import *  # noqa: F401,F403,E402
from hybrid_rag_system_amd.helper import main  # noqa: E402

if __name__ == "__main__":
    main()
