#!/bin/bash
# run the scan bench against debug builds of the library (experiments only)
set -e
cd $GRAFT_REPO_ROOT
cp hybrid_rag_system_amd/libhrs.so /tmp/libhrs.keep
for d in "$@"; do
  cp hybrid_rag_system_amd/libhrs_dbg$d.so hybrid_rag_system_amd/libhrs.so
  echo "== dbg $d"
  HRS_SCAN_Q2=1 timeout -k 10 100 python -u tools/mfma_bench.py 10000000x384x1024 | tail -1
done
cp /tmp/libhrs.keep hybrid_rag_system_amd/libhrs.so
