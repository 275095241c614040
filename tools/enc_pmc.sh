#!/bin/bash
# Vector-memory-path counters and phase clocks of the encoder GEMMs (run on the GPU box):
#   bash tools/enc_pmc.sh <tag>  ->  gpurun_out/<tag>_encoder_phases.txt
# Needs hybrid_rag_system_amd/libhrs_encprof.so (make -C hybrid_rag_system_amd/csrc encprof) for the phase clocks.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag="$1"
out=gpurun_out/${tag}_encoder_phases.txt
echo "== phase clocks inside k_gemm_ws<GELU> (HRS_LIBRARY=libhrs_encprof.so python tools/enc_phases.py)" > $out
HRS_LIBRARY=$PWD/hybrid_rag_system_amd/libhrs_encprof.so timeout -k 10 120 python3 tools/enc_phases.py >> $out 2>&1 || exit 1
for c in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
  echo "== rocprofv3 --pmc $c --kernel-trace --stats -- python3 tools/enc_bench.py (per launch; GRBM_GUI_ACTIVE / 8 / duration = clock)" >> $out
  n=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/encpmc_$n
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --stats -d gpurun_out/encpmc_$n --output-format csv -- python3 tools/enc_bench.py > gpurun_out/encpmc_$n.log 2>&1 || { echo "pass failed" >> $out; tail -3 gpurun_out/encpmc_$n.log >> $out; exit 1; }
  python3 tools/pmc_summary.py gpurun_out/encpmc_$n "k_gemm" >> $out
  python3 tools/kstats.py gpurun_out/encpmc_$n k_gemm >> $out
done
cat $out
