#!/bin/bash
# Held clock and duration of the dominant kernel for several builds of the library
# (build/ab/libdsv_<name>.so, "cur" = in-tree): one rocprofv3 PMC pass each, DSV_SPLIT=0 so that
# one dispatch = one whole 2^20 batch.  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration.
#   tools/clock_ab.sh OUTDIR name1 name2 ...
export TMPDIR=/tmp
export DSV_SPLIT=0
OUT=$1; shift
mkdir -p $OUT
for which in "$@"; do
  echo "[clock_ab] $which: PMC pass"
  if [ $which = cur ]; then unset DSV_LIB_PATH; else export DSV_LIB_PATH=$PWD/build/ab/libdsv_$which.so; fi
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$which -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-double > $OUT/$which.log 2>&1
  # FETCH_SIZE and WRITE_SIZE in SEPARATE passes (together the run never finishes on this pool)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "[clock_ab] $which: $ctr pass"
    timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${which}_$ctr -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-double > $OUT/${which}_$ctr.log 2>&1
  done
done
python3 - $OUT "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for which in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for sub in (which, which + "_FETCH_SIZE", which + "_WRITE_SIZE"):
        for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_verify_fixed_half" not in row["Kernel_Name"]:
                    continue
                d = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    agg["dur_ns"].append(d)
                    agg["clock_ghz"].append(float(row["Counter_Value"]) / 8.0 / d)
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    print("%-10s k_verify_fixed_half: %.3f ms  clock %.3f GHz  VALU wave-instr %.3e  FETCH %.2f GB  WRITE %.2f GB" % (
        which, m.get("dur_ns", 0) / 1e6, m.get("clock_ghz", 0), m.get("SQ_INSTS_VALU", 0),
        m.get("FETCH_SIZE", 0) / 1e6, m.get("WRITE_SIZE", 0) / 1e6))
PY
