#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).  Kernel trace + stats in one
# run, PMC counters in their own runs (never together with sys/hip traces).
#   trace        : the default command (2^16-item sub-batches on two streams)
#   trace_nosplit: DSV_SPLIT=0, one launch per kernel per step — the per-launch durations that
#                  bench.py's roofline.model.kernel_ms / hash_kernel_ms must agree with
#   pmc_*        : DSV_SPLIT=0 so that one dispatch = one whole 2^20 batch
# Summarise with tools/summarize_prof.py.
set -e
export TMPDIR=/tmp
OUT=${1:-gpurun_out/prof}
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-double"
mkdir -p $OUT
step() { echo "[profile_bench] $1"; }
step "rocprofv3 --kernel-trace --stats --output-format"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1   # incl. double / vargen / mixed kernels
step "export DSV_SPLIT=0"
export DSV_SPLIT=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_nosplit -- python3 $ARGS > $OUT/trace_nosplit.log 2>&1
step "rocprofv3 --pmc SQ_INSTS_VALU"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
step "rocprofv3 --pmc SQ_INSTS_VMEM_RD"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_IFETCH SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1 || true
step "rocprofv3 --pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
step "rocprofv3 --pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
# where the fetched bytes are served from (r05): L2 hit rate, requests that leave the L2 for the fabric
# (Infinity Cache / HBM) and their average latency (RDREQ_LEVEL / RDREQ: ~350 cycles = Infinity Cache,
# ~700 = HBM on an idle chip, /opt/skills/guides/MI355X_MICROARCH.md) — four TCC slots per pass
step "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 $ARGS > $OUT/pmc_tcc.log 2>&1 || true
step "rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc2 -- python3 $ARGS > $OUT/pmc_tcc2.log 2>&1 || true
# the fused double kernel (roofline_double.traffic): the same two counters over a run that includes the double leg
ARGS2="bench.py --steps 2 --warmup 1 --no-cpu-baseline"
step "rocprofv3 --pmc FETCH_SIZE (double leg)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch2 -- python3 $ARGS2 > $OUT/pmc_fetch2.log 2>&1
step "rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE (double leg)"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_write2 -- python3 $ARGS2 > $OUT/pmc_write2.log 2>&1
find $OUT -name "*.csv" | head -50
