#!/usr/bin/env python3
"""Per-kernel call count and average duration from the rocpd databases rocprofv3 writes by default
(`rocprofv3 --kernel-trace -d DIR -- ...`):  tools/rocpd_kernels.py DIR [substring]"""
import glob
import sqlite3
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    con = sqlite3.connect(f)
    for name, calls, total, avg, pct in con.execute("select * from top_kernels"):
        if pat in name:
            print("%-90s calls %5d  avg %10.3f us  %5.1f %%" % (name[:90], calls, avg, pct))
