#!/bin/bash
# PMC passes over the challenge-hash kernel for one build (DSV_LIB_PATH) with the two kernels of a
# step run back to back (DSV_SPLIT=0):  tools/prof_hash.sh OUTDIR
export TMPDIR=/tmp
export DSV_SPLIT=0
OUT=$1
mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-double > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv,glob,collections,sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/pmc*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:40]
        if "k_challenge" in k or "k_verify_fixed_half" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            agg[k]["dur_us"].append((int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e3)
for k,v in agg.items():
    print(k, {c:round(sum(x)/len(x)) for c,x in v.items()})
PY
