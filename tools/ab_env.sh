#!/bin/bash
# Same-box A/B of one build under two settings of an environment knob read at dsv_init:
#   tools/ab_env.sh ROUNDS VAR valueA valueB      e.g.  tools/ab_env.sh 3 DSV_DOUBLE_FUSED 0 1
ROUNDS=$1; VAR=$2; shift 2
for i in $(seq $ROUNDS); do
  for v in "$@"; do
    env $VAR=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('$VAR=$v', 'single', round(d['value']/1e6,2), 'double', round(d['double']['value']/1e6,2), 'vargen', round(d['vargen']['value']/1e6,2), 'mixed', round(d['mixed']['value']/1e6,2), 'M/s')"
  done
done
