set -e
for cfg in "" "DSV_PIPE_FIRST_LOG2=18" "DSV_PIPE_FIRST_LOG2=17" "DSV_HOST_THREADS=8" "DSV_PIPE_FIRST_LOG2=18 DSV_HOST_THREADS=8" "DSV_PIPE_FIRST_LOG2=16"; do
  env $cfg python bench.py --no-double 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=j['verify_batch_e2e']['fast_accept']
print('$cfg', 'fast all_valid %.2f ms (median %.2f) graded %.2f | one_shot %.2f' % (f['all_valid']['best_ms'], f['all_valid']['median_ms'], f['graded_workload']['best_ms'], j['verify_batch_e2e']['one_shot']['best_ms']))"
done
