#!/bin/bash
# run a tool against experimental builds of the library: tools/dbg_run.sh "<python cmd>" suffix...
set -e
cd $GRAFT_REPO_ROOT
cmd="$1"; shift
cp hybrid_rag_system_amd/libhrs.so /tmp/libhrs.keep
for d in "$@"; do
  cp hybrid_rag_system_amd/libhrs_$d.so hybrid_rag_system_amd/libhrs.so
  echo "== $d"
  timeout -k 10 200 $cmd
done
cp /tmp/libhrs.keep hybrid_rag_system_amd/libhrs.so
