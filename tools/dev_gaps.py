#!/usr/bin/env python3
"""Device-resident 2^20 single batch under rocprofv3 --kernel-trace: per hardware queue, the gaps between one
kernel's end and the next kernel's start inside a call (sub-batches alternate between two internal streams).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/dev_gaps.py run
    python3 tools/dev_gaps.py report DIR
"""
import csv
import glob
import os
import sys
import time


def run():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import torch
    from schnorr_amd import engine as E
    from schnorr_amd import workload as W
    E.init(0)
    n = 1 << 20
    b = W.gen_single(n, seed=2321)
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(E.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        E.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    torch.cuda.synchronize()
    time.sleep(0.05)
    t0 = time.perf_counter()
    for _ in range(4):
        E.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    torch.cuda.synchronize()
    print("4 calls: %.2f ms per call" % ((time.perf_counter() - t0) * 1e3 / 4))
    assert bool((ok == b["expected"]).all())


def report(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_challenge" in r["Kernel_Name"] or "k_verify_fixed_half" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][-24:]))
    rows.sort()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(x[1] for x in rows[max(0, i - 40):i]) > 30e6:
            cut = i
    call = rows[cut:]
    t0, t1 = call[0][0], max(r[1] for r in call)
    print("measured region: %d kernels, %.3f ms" % (len(call), (t1 - t0) / 1e6))
    byq = {}
    for s, e, q, k in call:
        byq.setdefault(q, []).append((s, e, k))
    for q, ks in sorted(byq.items()):
        gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
        busy = sum(e - s for s, e, _ in ks) / 1e6
        small = [g for g in gaps if g < 200]
        print("queue %s: %d kernels, busy %.3f ms, gaps: n=%d median %.1f us mean %.1f us sum %.3f ms (gaps >= 200 us: %s)" % (
            q, len(ks), busy, len(small), sorted(small)[len(small) // 2] if small else 0, sum(small) / max(1, len(small)),
            sum(small) / 1e3, ["%.0f" % g for g in gaps if g >= 200]))
    durs = {}
    for s, e, q, k in call:
        durs.setdefault(k, []).append((e - s) / 1e3)
    for k, v in durs.items():
        v.sort()
        print("%-26s n=%d median %.0f us min %.0f max %.0f" % (k, len(v), v[len(v) // 2], v[0], v[-1]))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
