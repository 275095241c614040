#!/bin/bash
# Effective clock and matrix-pipe occupancy of the headline's main pass, one --pmc pass each (run on the GPU box):
#   bash tools/pmc_clock.sh <tag>  ->  gpurun_out/<tag>_main_pass_clock.txt
# effective clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag="$1"
out=gpurun_out/${tag}_main_pass_clock.txt
: > $out
for c in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  echo "== $c" >> $out
  n=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmc_$n
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$n --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/pmc_$n.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_$n "k_mfma_scan16<12, 4, 64>" >> $out
  python3 - gpurun_out/pmc_$n >> $out <<'P'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))
if f:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f[-1])) if "k_mfma_scan16<12, 4, 64>" in r["Kernel_Name"]]
    d = [x for x in d if x > 1.0]
    print("   launch duration in this pass: n=%d avg=%.4f ms" % (len(d), sum(d) / max(len(d), 1)))
P
done
cat $out
