#!/bin/bash
# A/B of an environment switch on ONE box. usage: bash tools/ab_env.sh VAR [bench args]
V=$1; shift
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))" || exit 1
  env $V=1 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))" || exit 1
done
