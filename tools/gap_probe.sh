# Step time of the headline against the knobs of the sampled passes and of the re-score kernel (run on the GPU box).
set -e
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" HRS_STATS_VERBOSE=1 timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 2> gpurun_out/gap_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q/s %.0f  ms/step %.4f  main pass ms %.4f  longest list %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['max_candidates']))"; tail -1 gpurun_out/gap_err.txt; }
run A=1
run HRS_SS_MAX=64
run HRS_SS_MAX=64 HRS_J_MIN=8
run HRS_SS_MAX=48 HRS_J_MIN=12
run HRS_SS_MAX=64 HRS_J_MIN=8 HRS_SS2_MULT=1
run HRS_SS_MAX=16
run A=1
