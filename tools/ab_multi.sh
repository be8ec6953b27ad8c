#!/bin/bash
# Same-box A/B of several builds of the library (build/ab/libdsv_<name>.so; "cur" = in-tree),
# alternating so that clock drift hits all alike:  tools/ab_multi.sh ROUNDS name1 name2 ...
set -e
ROUNDS=$1; shift
for i in $(seq $ROUNDS); do
  for which in "$@"; do
    if [ $which = cur ]; then unset DSV_LIB_PATH; else export DSV_LIB_PATH=$PWD/build/ab/libdsv_$which.so; fi
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['roofline'].get('model',{})
print('%-10s' % '$which', round(d['value']/1e6,2), 'M/s  step', round(d['ms_per_step'],3), 'verify', round(m.get('kernel_ms',0),3), 'hash', round(m.get('hash_kernel_ms',0),3), 'double', round(d.get('double',{}).get('value',0)/1e6,2), 'vargen', round(d.get('vargen',{}).get('value',0)/1e6,2))"
  done
done
