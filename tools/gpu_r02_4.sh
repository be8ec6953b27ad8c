#!/bin/bash
set -e
mkdir -p gpurun_out
bash tools/profile_bench.sh gpurun_out/prof_r02 > gpurun_out/r02_4_prof.log 2>&1 || { tail -30 gpurun_out/r02_4_prof.log; exit 1; }
python3 tools/summarize_prof.py gpurun_out/prof_r02 gpurun_out/r02v1 > gpurun_out/r02_4_sum.log 2>&1 || tail -20 gpurun_out/r02_4_sum.log
cp profiles/pmc_latest.json gpurun_out/pmc_latest_r02.json
bash tools/clock_ab.sh gpurun_out/clock_ab cur storedneg grid2048 cur > gpurun_out/r02_4_clock_ab.txt 2>&1 || true
cat gpurun_out/r02_4_clock_ab.txt
bash tools/ab_multi.sh 3 storedneg cur grid2048 > gpurun_out/r02_4_ab.txt 2>&1 || true
cat gpurun_out/r02_4_ab.txt
rm -rf gpurun_out/prof_r02/*/*/*.db gpurun_out/clock_ab/*/*/*.db 2>/dev/null || true
du -sh gpurun_out/prof_r02 gpurun_out/clock_ab
