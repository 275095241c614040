#!/bin/bash
# HBM fetch / write bytes per kernel against experimental builds: tools/dbg_pmc.sh "<python3 script args>" <kernel filter> suffix...
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cmd="$1"; filt="$2"; shift; shift
cp hybrid_rag_system_amd/libhrs.so /tmp/libhrs.keep
for d in "$@"; do
  cp hybrid_rag_system_amd/libhrs_$d.so hybrid_rag_system_amd/libhrs.so
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $d $c"
    rm -rf gpurun_out/pmc_dbg
    timeout -s KILL 120 rocprofv3 --pmc $c -d gpurun_out/pmc_dbg --output-format csv -- python3 $cmd > gpurun_out/dbg.log 2>&1 || true
    python3 tools/pmc_summary.py gpurun_out/pmc_dbg "$filt"
  done
done
cp /tmp/libhrs.keep hybrid_rag_system_amd/libhrs.so
