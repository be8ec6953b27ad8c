#!/bin/bash
# Samples rocm-smi (power, clocks, temperature) while the verify path runs, to tell whether the
# sustained shader clock under this VALU load is power- or thermally limited.
set -e
OUT=${1:-gpurun_out/clock_probe.txt}
python3 bench.py --steps 900 --warmup 2 --no-cpu-baseline --no-double > gpurun_out/clock_probe_bench.json 2>/dev/null &
BPID=$!
sleep 9
for i in 1 2 3 4 5 6; do
  echo "== sample $i" >> $OUT
  rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -v "^=\|^$" >> $OUT || true
  sleep 0.4
done
wait $BPID
tail -c 400 gpurun_out/clock_probe_bench.json >> $OUT
