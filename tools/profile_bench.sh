#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).  Kernel trace + stats in one
# run, PMC counters in their own runs (never together with sys/hip traces).
set -e
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-double"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_IFETCH SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
