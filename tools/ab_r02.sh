#!/bin/bash
# Same-box A/B of the round-2 final tree (build/r02tree: `git archive c23c5db` + its library, prepared
# on the build host; build/ is git-ignored but travels with gpurun) against the working tree,
# alternating runs:  tools/ab_r02.sh [ROUNDS]
set -e
ROUNDS=${1:-3}
ROOT=$PWD
for i in $(seq $ROUNDS); do
  for which in r02 cur; do
    if [ $which = r02 ]; then cd $ROOT/build/r02tree; else cd $ROOT; fi
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']
ms=lambda pat: next((round(v['ms_per_launch'],3) for n,v in k.items() if pat in n), None)
print('%-4s' % '$which', 'single', round(d['value']/1e6,2), 'double', round(d['double']['value']/1e6,2), 'vargen', round(d['vargen']['value']/1e6,2),
      'mixed', round(d['mixed']['value']/1e6,2), '| ms: verify1', ms('half<false,1>') or ms('half<1>'), 'verify2', ms('half<false,2>') or ms('half<2>'),
      'hash1', ms('challenge<false>'), 'hash2', ms('challenge<true>'), 'var', ms('verify_var'))"
  done
done
