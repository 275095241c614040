#!/bin/bash
# config 1 through the drop-in CLI: wall time of one-shot invocations (process start -> output)
set -e
cd $GRAFT_REPO_ROOT
export HRS_DATA_DIR=$(mktemp -d)
t() { local s=$(date +%s%N); "$@" > /tmp/cli_out.txt 2>/tmp/cli_err.txt; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms : $*"; }
t python scripts/chromadb_helper.py index tests/golden/context/infrastructure -r
tail -2 /tmp/cli_out.txt
t python scripts/chromadb_helper.py query "How do I restart the nginx container on Unraid?" --hybrid -n 5 --json
head -c 300 /tmp/cli_out.txt; echo
t python scripts/chromadb_helper.py query "powershell remoting" -n 5 --json
t python scripts/mcp_chromadb_bridge.py query "docker compose" --hybrid --n-results 5
head -c 200 /tmp/cli_out.txt; echo
