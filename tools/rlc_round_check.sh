#!/bin/bash
# closing run for the batch fast accept: GPU suite, smoke, soak, probe sweep, kernel trace, bench line
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_suite.log 2>&1
python __graft_entry__.py smoke > gpurun_out/r05_smoke.log 2>&1
timeout -k 10 900 python tools/soak_rlc.py 240 11 > gpurun_out/soak_rlc.txt 2>&1
{
for a in "20 0 8 single" "22 0 5 single" "18 0 8 single" "16 0 8 single" "20 0 6 double" "18 0 8 double" "20 0 6 vargen" "18 0 8 vargen"; do
  timeout -k 10 200 python tools/rlc_probe.py $a 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/rlc_probe_final.txt
rocprofv3 --kernel-trace -d gpurun_out/rlc_prof -o rlc -- python3 tools/rlc_probe.py 20 0 4 > gpurun_out/rlc_prof.log 2>&1
python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
