import csv, glob, sys
d=sys.argv[1]
rows=[]
for f in glob.glob(d+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("dsv::","")[-34:], r["Queue_Id"]))
for f in glob.glob(d+"/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r["Direction"][-16:], "-"))
rows.sort()
# take a window of ~3 host calls in the middle of the host-call phase
hs=[r for r in rows if "k_verify_fixed_half_oct" in r[2]]
mid=hs[len(hs)//4][0]
w=[r for r in rows if mid-50_000 <= r[0] <= mid+1_100_000]
t0=w[0][0]
for s,e,k,q in w:
    print("%9.1f %9.1f %7.1f us  %-36s q=%s"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,k,q))
