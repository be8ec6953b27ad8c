#!/bin/bash
# Same-box A/B of two builds of the library: $1 (default build/ab/libdsv_prev.so, built from
# another commit by hand; build/ is git-ignored but travels with gpurun) against the in-tree one,
# alternating runs so clock drift affects both alike.
set -e
PREV=${1:-build/ab/libdsv_prev.so}
ROUNDS=${2:-3}
for i in $(seq $ROUNDS); do
  for which in prev cur; do
    if [ $which = prev ]; then export DSV_LIB_PATH=$PWD/$PREV; else unset DSV_LIB_PATH; fi
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['roofline'].get('model',{})
print('$which', round(d['value']/1e6,2), 'M/s  step', round(d['ms_per_step'],3), 'verify', round(m.get('kernel_ms',0),3), 'hash', round(m.get('hash_kernel_ms',0),3), 'double', round(d.get('double',{}).get('value',0)/1e6,2), 'vargen', round(d.get('vargen',{}).get('value',0)/1e6,2))"
  done
done
