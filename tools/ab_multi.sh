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
k=d['roofline']['kernels']
ms=lambda pat: next((round(v['ms_per_launch'],3) for n,v in k.items() if pat in n), None)
g=lambda key: round(d.get(key,{}).get('value',0)/1e6,2)
print('%-10s' % '$which', 'single', round(d['value']/1e6,2), 'double', g('double'), 'vargen', g('vargen'), 'mixed', g('mixed'), 'ext', g('ext'), 'wire', g('wire'),
      '| ms: verify1', ms('half<1>'), 'verify2', ms('half<2>'), 'hash1', ms('challenge<false>'), 'hash2', ms('challenge<true>'), 'var', ms('verify_var'))"
  done
done
