#!/bin/bash
# Wave-state and LDS counters of the encoder kernels, one --pmc pass each (run on the GPU box):
#   bash tools/enc_sq_pmc.sh <tag>  ->  gpurun_out/<tag>_encoder_sq.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag="$1"
out=gpurun_out/${tag}_encoder_sq.txt
: > $out
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  echo "== rocprofv3 --pmc $c --kernel-trace --stats -- python3 tools/enc_bench.py (sums over all waves / SQs, per launch)" >> $out
  n=$(echo $c | tr ' ' '_' | cut -c1-60)
  rm -rf gpurun_out/encsq_$n
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --stats -d gpurun_out/encsq_$n --output-format csv -- python3 tools/enc_bench.py > gpurun_out/encsq_$n.log 2>&1 || { echo "pass failed" >> $out; tail -3 gpurun_out/encsq_$n.log >> $out; exit 1; }
  python3 tools/pmc_summary.py gpurun_out/encsq_$n "k_gemm" >> $out
  python3 tools/pmc_summary.py gpurun_out/encsq_$n "k_attention" >> $out
done
cat $out
