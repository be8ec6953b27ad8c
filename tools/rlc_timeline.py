#!/usr/bin/env python3
"""Timeline of ONE fast-accept call from a rocprofv3 --kernel-trace database:

    rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python3 tools/rlc_case.py 20 single valid 0 4
    python tools/rlc_timeline.py gpurun_out/tl/tl_results.db

prints every kernel between the last two k_rlc_verdict launches (= the last call) in start order: offset
from the call's first kernel, duration, gap to the previous kernel's end, grid."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
ends = [i for i, r in enumerate(rows) if "k_rlc_verdict" in r[0]]
if len(ends) < 2:
    sys.exit("need at least two calls in the trace")
call = rows[ends[-2] + 1:ends[-1] + 1]
t0 = call[0][1]
last_end = t0
print("# %d kernels, %.3f ms from the first kernel's start to the verdict's end" % (len(call), (call[-1][2] - t0) / 1e6))
for name, s, e, gx, gy, wx in call:
    short = name.split("(")[0].replace("void ", "").replace("dsv::", "")
    print("%9.1f us  +%8.1f us  gap %7.1f  %-44s grid %d x %d / %d" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3,
                                                                         short[:44], gx // max(wx, 1), gy, wx))
    last_end = max(last_end, e)
