#!/bin/bash
# profile a tool against experimental builds of the library: tools/dbg_prof.sh "<python3 script args>" <kernel filter> suffix...
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cmd="$1"; filt="$2"; shift; shift
cp hybrid_rag_system_amd/libhrs.so /tmp/libhrs.keep
for d in "$@"; do
  cp hybrid_rag_system_amd/libhrs_$d.so hybrid_rag_system_amd/libhrs.so
  echo "== $d"
  rm -rf gpurun_out/prof_dbg
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dbg --output-format csv -- python3 $cmd > gpurun_out/dbg.log 2>&1 || true
  tail -1 gpurun_out/dbg.log
  python3 tools/kstats.py gpurun_out/prof_dbg $filt
done
cp /tmp/libhrs.keep hybrid_rag_system_amd/libhrs.so
