#!/usr/bin/env python3
"""`python3 mcp_chromadb_bridge.py query ...` as the MCP server spawns it: see hybrid_rag_system_amd.bridge."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from hybrid_rag_system_amd.bridge import format_hybrid_results, format_query_results, main  # noqa: F401,E402

if __name__ == "__main__":
    main()
