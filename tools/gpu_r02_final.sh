#!/bin/bash
# r02 closing run: full GPU test suite, rocprofv3 passes of the final build, final bench line.
mkdir -p gpurun_out
echo "[final] tests"
python3 -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/r02_final_tests.log | tail -4
echo "[final] profiles"
rm -rf gpurun_out/prof_r02f
bash tools/profile_bench.sh gpurun_out/prof_r02f 2>&1 | grep "profile_bench"
python3 tools/summarize_prof.py gpurun_out/prof_r02f gpurun_out/r02v2 > gpurun_out/r02_final_sum.log 2>&1 || tail -5 gpurun_out/r02_final_sum.log
cp profiles/pmc_latest.json gpurun_out/pmc_latest_r02v2.json
find gpurun_out/prof_r02f -name "*.db" -delete 2>/dev/null
echo "[final] bench"
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err
cat gpurun_out/r02_final_bench.json | cut -c1-600
