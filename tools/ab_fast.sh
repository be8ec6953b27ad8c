#!/bin/bash
# bench.py's verify_batch_e2e.fast_accept (typed objects -> verify_batch_fast) under several settings
set -e
for cfg in "" "DSV_MULTI_SHARDS=2" "DSV_MULTI_SHARDS=3" "DSV_MULTI_SHARDS=4" "DSV_MULTI_SHARDS=2 DSV_HOST_THREADS=8"; do
  env $cfg python bench.py --no-double 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=j['verify_batch_e2e']['fast_accept']
print('$cfg', 'fast all_valid %.2f ms (median %.2f) two-in-flight %.2f graded %.2f | one_shot %.2f' % (f['all_valid']['best_ms'], f['all_valid']['median_ms'], f.get('all_valid_two_in_flight',{}).get('ms_per_call',0), f['graded_workload']['best_ms'], j['verify_batch_e2e']['one_shot']['best_ms']))"
done
