#!/bin/bash
# r02 closing run: full GPU test suite, rocprofv3 passes of the final build, final bench line.
mkdir -p gpurun_out
echo "[final] tests"
python3 -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/round_tests.log | tail -4
echo "[final] profiles"
rm -rf gpurun_out/prof_round
bash tools/profile_bench.sh gpurun_out/prof_round 2>&1 | grep "profile_bench"
python3 tools/summarize_prof.py gpurun_out/prof_round gpurun_out/round > gpurun_out/round_sum.log 2>&1 || tail -5 gpurun_out/round_sum.log
cp profiles/pmc_latest.json gpurun_out/pmc_latest_round.json
find gpurun_out/prof_round -name "*.db" -delete 2>/dev/null
echo "[final] bench"
python3 bench.py --steps 20 --warmup 3 > gpurun_out/round_bench.json 2> gpurun_out/round_bench.err
cat gpurun_out/round_bench.json | cut -c1-600
