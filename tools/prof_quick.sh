#!/bin/bash
export TMPDIR=/tmp
OUT=$1
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-double > $OUT/pmc.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/pmc/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    k=row["Kernel_Name"][:44]
    if "dsv::" in k:
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        agg[k]["dur_us"].append((int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e3)
for k,v in agg.items():
    print(k, {c:round(sum(x)/len(x)) for c,x in v.items()})
PY
