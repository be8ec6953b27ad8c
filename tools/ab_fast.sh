#!/bin/bash
# bench.py's verify_batch_e2e.fast_accept (typed objects -> verify_batch_fast) and the host wire form under several settings
set -e
for cfg in "DSV_RLC_STAGED=1" "DSV_RLC_STAGED=0" "DSV_RLC_STAGED=1" "DSV_RLC_STAGED=0"; do
  env $cfg python bench.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=j['verify_batch_e2e']['fast_accept']
w=j['wire']['host'].get('fast_accept_all_valid',{})
print('$cfg', 'typed objects: all_valid %.2f ms (median %.2f) two-in-flight %.2f graded %.2f | wire host %.2f ms | verify_batch one_shot %.2f' % (f['all_valid']['best_ms'], f['all_valid']['median_ms'], f.get('all_valid_two_in_flight',{}).get('ms_per_call',0), f['graded_workload']['best_ms'], w.get('ms_per_call',0), j['verify_batch_e2e']['one_shot']['best_ms']))"
done
