#!/bin/bash
mkdir -p gpurun_out
echo "[r02_5] clock A/B"
timeout -k 10 900 bash tools/clock_ab.sh gpurun_out/clock_ab cur storedneg grid2048 cur 2>&1 | tee gpurun_out/r02_5_clock_ab.txt
echo "[r02_5] throughput A/B"
bash tools/ab_multi.sh 3 storedneg cur grid2048 2>&1 | tee gpurun_out/r02_5_ab.txt
find gpurun_out/clock_ab -name "*.db" -delete 2>/dev/null
du -sh gpurun_out/clock_ab
