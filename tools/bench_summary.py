#!/usr/bin/env python3
"""One-line-per-figure summary of one or more bench.py JSON lines (files given on the command line)."""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    top = {k: (round(d[k]["value"] / 1e6, 2) if isinstance(d.get(k), dict) and "value" in d[k] else None)
           for k in ("double", "vargen", "mixed", "sign", "host_path", "host_path_ext", "wire")}
    print(path)
    print("  single %.2f M/s (%.3f ms/step)  %s" % (d["value"] / 1e6, d["ms_per_step"],
                                                   "  ".join("%s %s" % kv for kv in top.items() if kv[1] is not None)))
    if "small_batch" in d:
        print("  small batch %.3f ms" % d["small_batch"]["ms_per_call"])
    for k, v in d.get("roofline", {}).get("kernels", {}).items():
        print("  %-62s %8.3f ms  mad_frac %.3f" % (k[:62], v["ms_per_launch"], v["mad_frac"]))
