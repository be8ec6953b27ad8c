#!/bin/bash
set -e
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_3_tests.log 2>&1 || { tail -60 gpurun_out/r02_3_tests.log; exit 1; }
tail -3 gpurun_out/r02_3_tests.log
bash tools/ab_multi.sh 2 noneg cur > gpurun_out/r02_3_ab.txt 2>&1 || true
cat gpurun_out/r02_3_ab.txt
python3 bench.py --steps 10 --warmup 2 > gpurun_out/r02_3_bench.json 2> gpurun_out/r02_3_bench.err || { tail -20 gpurun_out/r02_3_bench.err; exit 1; }
cat gpurun_out/r02_3_bench.json
