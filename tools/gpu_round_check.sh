#!/bin/bash
# Closing run of a round on the GPU box: full GPU test suite, rocprofv3 passes of the final build
# (kernel trace + stats, then PMC passes on their own), summaries for profiles/<round>/, final bench
# line.  Everything lands under gpurun_out/ (progress lines on stdout as it goes).
#   tools/gpu_round_check.sh r03
#   (a failing or timed-out step ends the run: no further GPU step is started after it)
R=${1:-r03}
mkdir -p gpurun_out
echo "[final] tests"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/${R}_round_tests.log 2>&1 || { tail -15 gpurun_out/${R}_round_tests.log; exit 1; }
tail -3 gpurun_out/${R}_round_tests.log
# the Rust shim's own tests, where a toolchain and the crates exist (they were never run)
if command -v cargo >/dev/null 2>&1; then
  echo "[final] cargo test (rust/dusk-schnorr-gpu)"
  (cd rust/dusk-schnorr-gpu && timeout -k 10 900 cargo test --release --offline) > gpurun_out/${R}_cargo_test.log 2>&1 \
    && tail -3 gpurun_out/${R}_cargo_test.log || { echo "[final] cargo test FAILED (or no registry)"; tail -5 gpurun_out/${R}_cargo_test.log; }
else
  echo "[final] no cargo on this box: the Rust shim stays uncompiled (DESIGN.md, parity status)"
fi
echo "[final] profiles"
rm -rf gpurun_out/prof_round
bash tools/profile_bench.sh gpurun_out/prof_round 2>&1 | grep "profile_bench" || exit 1
python3 tools/summarize_prof.py gpurun_out/prof_round gpurun_out/${R}_final > gpurun_out/${R}_round_sum.log 2>&1 || tail -5 gpurun_out/${R}_round_sum.log
cp profiles/pmc_latest.json gpurun_out/${R}_pmc_latest.json
find gpurun_out/prof_round -name "*.db" -delete 2>/dev/null
echo "[final] bench"
python3 bench.py --steps 20 --warmup 3 > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.err
python3 tools/bench_summary.py gpurun_out/${R}_bench_final.json
