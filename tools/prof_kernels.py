#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace result database (rocpd sqlite):
    python tools/prof_kernels.py gpurun_out/x/x_results.db [name filter]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
flt = "%" + (sys.argv[2] if len(sys.argv) > 2 else "") + "%"
rows = list(cur.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, sum(end-start)/1e6, max(grid_x), max(workgroup_x) "
                        "from kernels where name like ? group by name order by 5 desc", (flt,)))
for r in rows[:40]:
    print("%-72s n=%5d avg %9.1f us min %9.1f us total %9.2f ms grid %d x %d" % (r[0][:72], r[1], r[2], r[3], r[4], r[5], r[6]))
