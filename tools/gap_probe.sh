set -e
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" HRS_STATS_VERBOSE=1 timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 2> gpurun_out/gap_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['max_candidates'])"; tail -1 gpurun_out/gap_err.txt; }
run A=1
run HRS_SEL_FIRST=128
run HRS_SEL_FIRST=192
run HRS_SS2_MULT=4
run HRS_SS2_MULT=4 HRS_SEL_FIRST=192
run HRS_PER_K=12
run HRS_PER_K=12 HRS_SEL_FIRST=192
run A=1
