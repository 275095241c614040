#!/bin/bash
# A/B on ONE box: the headline bench (no extras) with two builds of the library, alternating.  usage: bash tools/ab.sh libA.so libB.so [bench args]
A=$1; B=$2; shift 2
for i in 1 2; do
  for L in $A $B; do
    HRS_LIBRARY=$PWD/$L timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))" || exit 1
  done
done
