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
    e = d.get("verify_batch_e2e")
    if e:
        s = e.get("streamed", {})
        print("  verify_batch (typed objects): one shot %.2f ms = %.1f M/s, two in flight %.2f ms = %.1f M/s" % (
            e["one_shot"]["best_ms"], e["one_shot"]["value"] / 1e6,
            s.get("two_in_flight", {}).get("ms_per_call", 0), s.get("two_in_flight", {}).get("value", 0) / 1e6))
    r = d.get("rlc")
    if r:
        f = (e or {}).get("fast_accept", {})
        w = d.get("wire", {})
        print("  fast accept, all-valid batch: single %.1f M/s (x%.2f)  double %.1f  vargen %.1f | graded workload %.1f M/s (x%.2f)" % (
            r["all_valid"]["value"] / 1e6, r["all_valid"]["vs_per_signature"],
            r.get("double_all_valid", {}).get("value", 0) / 1e6, r.get("vargen_all_valid", {}).get("value", 0) / 1e6,
            r["graded_workload"]["value"] / 1e6, r["graded_workload"]["vs_per_signature"]))
        ob = lambda k: "%.2f ms x%.2f" % (r[k]["ms_per_call"], r[k]["vs_per_signature"]) if k in r else "-"
        print("    one wrong signature in the batch: failed within 8 calls %s | within 128 calls (guarded) %s | never before %s | two streams, all valid %.1f M/s" % (
            ob("one_bad_in_batch"), ob("one_bad_first_in_a_while"), ob("one_bad_never_seen_before"),
            r["all_valid"].get("two_streams", {}).get("value", 0) / 1e6))
        print("    typed objects %.1f M/s (two in flight %.1f)  wire records %.1f M/s in HBM, %.1f M/s from host memory" % (
            f.get("all_valid", {}).get("value", 0) / 1e6, f.get("all_valid_two_in_flight", {}).get("value", 0) / 1e6,
            w.get("fast_accept_all_valid", {}).get("value", 0) / 1e6,
            w.get("host", {}).get("fast_accept_all_valid", {}).get("value", 0) / 1e6))
