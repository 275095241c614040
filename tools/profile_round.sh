#!/bin/bash
# Kernel-trace profiles of the bench and of the encoder alone; summaries land in gpurun_out/ for copying into profiles/.
# usage (on the GPU box): bash tools/profile_round.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag="$1"
rm -rf gpurun_out/prof_bench gpurun_out/prof_enc
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${tag}_profiled_run.json 2> gpurun_out/${tag}_profiled_run.err
cp $(ls gpurun_out/prof_bench/*/*kernel_stats.csv | tail -1) gpurun_out/${tag}_kernel_stats.csv
python3 - "$tag" <<'P'
import csv, glob, sys
tag = sys.argv[1]
f = sorted(glob.glob("gpurun_out/prof_bench/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in rows if "k_mfma_scan16<12, 4, 64>" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1_000_000]
with open(f"gpurun_out/{tag}_main_kernel_launches.txt", "w") as o:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in main]
    o.write("k_mfma_scan16<12, 4, 64> (main pass, int8, v_mfma_i32_16x16x64_i8) launches over the 10M-row corpus, ms, in order:\n" + " ".join("%.3f" % x for x in d) + "\n")
    if len(d) >= 10:
        o.write("mean of the last ten (the timed steps): %.3f ms\n" % (sum(d[-10:]) / 10))
P
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_enc --output-format csv -- python3 tools/enc_bench.py > gpurun_out/${tag}_encoder_run.txt 2>&1
python3 tools/kstats.py gpurun_out/prof_enc k_ > gpurun_out/${tag}_encoder_kernels.txt
cat gpurun_out/${tag}_main_kernel_launches.txt gpurun_out/${tag}_encoder_kernels.txt
