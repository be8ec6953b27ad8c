#!/usr/bin/env python3
"""Condense the output of tools/profile_bench.sh into the small files kept under profiles/.

    python tools/summarize_prof.py gpurun_out/prof profiles/r01/v19

writes  <prefix>_kernel_stats.csv          rocprofv3 --stats table of the default command
        <prefix>_kernel_stats_nosplit.csv  same with DSV_SPLIT=0 (one launch per kernel per step)
        <prefix>_pmc_summary.json          per kernel: mean counter value per dispatch
and refreshes profiles/pmc_latest.json (what bench.py's roofline.traffic / valu_busy_from_pmc
read) from the k_verify_fixed_half rows.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    for sub, suffix in (("trace", ""), ("trace_nosplit", "_nosplit")):
        hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if hits:  # keep the engine's kernels and the runtime copies; torch's helper kernels are noise
            with open(hits[0]) as f, open("%s_kernel_stats%s.csv" % (prefix, suffix), "w") as g:
                for i, line in enumerate(f):
                    if i == 0 or "dsv::" in line or "__amd_rocclr" in line:
                        g.write(line)
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    clk = defaultdict(list)   # GRBM_GUI_ACTIVE / 8 XCDs / duration of the SAME dispatch = held GHz
    for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if not k.startswith("dsv::"):
                    continue
                # the passes over the double leg also run the mixed leg (2^19-item launches of both
                # verify kernels) and further single-signature legs: keep only the fused double kernel's
                # whole-batch launches from them (the half-size ones take half as long)
                if "pmc_fetch2" in path or "pmc_write2" in path:
                    if "k_verify_fixed_half<2>" not in k:
                        continue
                    d_ns = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    if d_ns < 15e6:
                        continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    d_ns = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    if d_ns > 0:
                        clk[k].append(float(row["Counter_Value"]) / 8.0 / d_ns)
                key = (path, row["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    out = {}
    for k, counters in sorted(acc.items()):
        out[k] = {c: {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
                  for c, v in sorted(counters.items())}
        out[k]["avg_duration_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
        if clk[k]:
            out[k]["clock_held_ghz"] = sum(clk[k]) / len(clk[k])
    with open(prefix + "_pmc_summary.json", "w") as f:
        json.dump(out, f, indent=1)
    dom = [k for k in out if k.startswith("dsv::k_verify_fixed_half<")]
    if dom:
        stats = {}
        nos = prefix + "_kernel_stats_nosplit.csv"
        if os.path.exists(nos):
            with open(nos) as f:
                for row in csv.DictReader(f):
                    if "k_verify_fixed_half<1>" in row["Name"]:
                        stats = row
        import datetime
        single = [k for k in dom if "<1>" in k] or dom
        d = out[single[0]]
        latest = {
            "source": "tools/profile_bench.sh + tools/summarize_prof.py -> %s_*; DSV_SPLIT=0 passes, "
                      "per launch over 2^20 signatures" % os.path.relpath(prefix, ROOT),
            # provenance (bench.py: roofline.pmc_source): when, which build, which command
            "captured": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"),
            "commit": os.environ.get("DSV_COMMIT"),
            "command": "DSV_SPLIT=0 rocprofv3 --pmc <one group per pass> --kernel-trace -- python3 bench.py "
                       "--steps 3 --warmup 1 --no-cpu-baseline --no-double (tools/profile_bench.sh)",
            "kernel": single[0],
            "batch": 1 << 20,
        }
        dbl = [k for k in dom if "<2>" in k]
        if dbl and "FETCH_SIZE" in out[dbl[0]] and "WRITE_SIZE" in out[dbl[0]]:
            dd = out[dbl[0]]
            latest["double"] = {"kernel": dbl[0], "FETCH_SIZE_KB": dd["FETCH_SIZE"]["avg_per_launch"],
                                "WRITE_SIZE_KB": dd["WRITE_SIZE"]["avg_per_launch"],
                                "avg_duration_ns_under_pmc": dd["avg_duration_ns_under_pmc"],
                                "clock_held_ghz": dd.get("clock_held_ghz")}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c in d:
                latest[c + "_KB"] = d[c]["avg_per_launch"]
        for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
            if c in d:
                latest[c] = d[c]["avg_per_launch"]
        # r05: what the recorded figures are valid for — bench.py marks them stale when the kernel's
        # sources (k_verify.hip and every header it includes) differ from these
        sys.path.insert(0, ROOT)
        from schnorr_amd import build as B
        latest["evidence"] = B.evidence_hashes()
        # r05: where the fetched bytes are served from
        tcc = {c: d[c]["avg_per_launch"] for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum",
                                                   "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_32B_sum",
                                                   "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum",
                                                   "TCC_EA0_WRREQ_64B_sum") if c in d}
        if "TCC_HIT_sum" in tcc and "TCC_MISS_sum" in tcc:
            tcc["l2_hit_rate"] = tcc["TCC_HIT_sum"] / max(1.0, tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"])
        if tcc.get("TCC_EA0_RDREQ_sum"):
            if "TCC_EA0_RDREQ_LEVEL_sum" in tcc:
                tcc["avg_fabric_read_latency_cycles"] = tcc["TCC_EA0_RDREQ_LEVEL_sum"] / tcc["TCC_EA0_RDREQ_sum"]
            if "TCC_EA0_RDREQ_32B_sum" in tcc:
                r32 = tcc["TCC_EA0_RDREQ_32B_sum"]
                tcc["fabric_read_bytes"] = 32.0 * r32 + 64.0 * (tcc["TCC_EA0_RDREQ_sum"] - r32)
        if tcc:
            latest["cache"] = tcc
        latest["avg_duration_ns_under_pmc"] = d["avg_duration_ns_under_pmc"]
        if "clock_held_ghz" in d:
            latest["clock_held_ghz"] = d["clock_held_ghz"]
        if stats:
            latest["avg_duration_ns"] = float(stats["AverageNs"])
        hk = [k for k in out if k.startswith("dsv::k_challenge<false>")]
        if hk:  # the other kernel of a step (bench.py adds its HBM bytes to roofline.traffic)
            h = out[hk[0]]
            latest["k_challenge"] = {c + ("_KB" if c.endswith("_SIZE") else ""): h[c]["avg_per_launch"]
                                     for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU") if c in h}
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as f:
            json.dump(latest, f, indent=1)
    print(json.dumps({k: {c: v["avg_per_launch"] for c, v in cs.items() if isinstance(v, dict)}
                      for k, cs in out.items()}, indent=1)[:3000])


if __name__ == "__main__":
    main()
