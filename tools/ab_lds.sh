#!/bin/bash
# A/B: fixed-base table staged in LDS (builds with -DDSV_FIXED_LDS_BITS=5 / 6, DSV_FIXED_LDS=1)
# against the shipped 11-bit table in L2.  tools/ab_lds.sh ROUNDS
for i in $(seq ${1:-2}); do
  for which in cur lds5 lds6; do
    if [ $which = cur ]; then unset DSV_LIB_PATH DSV_FIXED_LDS; else export DSV_LIB_PATH=$PWD/build/ab/libdsv_$which.so DSV_FIXED_LDS=1; fi
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-double 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['roofline']['model']
print('%-6s' % '$which', round(d['value']/1e6,2), 'M/s  step', round(d['ms_per_step'],3), 'verify kernel', round(m['kernel_ms'],3), 'hash', round(m['hash_kernel_ms'],3))"
  done
done
