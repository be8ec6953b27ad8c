#!/bin/bash
# r02 GPU call 1: parity tests of the FIPS multiplier build, same-box A/B against the r01 library,
# extended VALU-rate microbenchmark.
set -e
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/r02_1_tests.log 2>&1 || { tail -30 gpurun_out/r02_1_tests.log; exit 1; }
tail -3 gpurun_out/r02_1_tests.log
bash tools/ab_bench.sh build/ab/libdsv_r01.so 3 > gpurun_out/r02_1_ab.txt 2>&1
cat gpurun_out/r02_1_ab.txt
./tools/microbench/valu_rates > gpurun_out/r02_valu_rates.txt 2>&1
grep -E "and_b32|or_b32|sub_u32|lshrrev|not_b32|mov_b32|bitop3|nop|add_u32 |mad_u64_u32_sgpr|lshl_add_u64" gpurun_out/r02_valu_rates.txt | head -40
