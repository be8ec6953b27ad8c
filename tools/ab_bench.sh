#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab/libdsv_prev.so (built from another commit by
# hand) against the in-tree one, alternating runs so clock drift affects both alike.
set -e
for i in 1 2 3; do
  for which in prev cur; do
    if [ $which = prev ]; then export DSV_LIB_PATH=$PWD/tools/ab/libdsv_prev.so; else unset DSV_LIB_PATH; fi
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', round(d['value']/1e6,2), 'M/s  step', round(d['ms_per_step'],3), 'verify', round(d['roofline']['model']['kernel_ms'],3), 'hash', round(d['roofline']['model']['hash_kernel_ms'],3), 'double', round(d['double']['value']/1e6,2), 'vargen', round(d['vargen']['value']/1e6,2))"
  done
done
