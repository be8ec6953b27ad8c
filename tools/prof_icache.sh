#!/bin/bash
export TMPDIR=/tmp
OUT=$1
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -iE "ICACHE|SQC_INST|SQ_INSTS_VALU |SQ_INST_CYCLES|SQ_WAIT_INST" | head -30 > $OUT/list.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,collections,sys
fs=glob.glob(sys.argv[1]+"/pmc/*/*_counter_collection.csv")
if not fs: print(open(sys.argv[1]+"/pmc.log").read()[-1500:]); sys.exit()
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(fs[0])):
    k=row["Kernel_Name"][:44]
    if "dsv::" in k: agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items(): print(k, {c:round(sum(x)/len(x)) for c,x in v.items()})
PY
