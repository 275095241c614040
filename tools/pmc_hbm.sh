#!/bin/bash
# HBM fetch / write counters of the headline main pass, one --pmc pass each (run on the GPU box):
#   bash tools/pmc_hbm.sh <tag>  ->  gpurun_out/<tag>_bench_pmc_hbm.txt
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag="$1"
out=gpurun_out/${tag}_bench_pmc_hbm.txt
: > $out
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== $c" >> $out
  rm -rf gpurun_out/pmc_$c
  timeout -s KILL 120 rocprofv3 --pmc $c -d gpurun_out/pmc_$c --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_$c "k_mfma_scan16<12, 4, 64>" >> $out
done
cat $out
