#!/usr/bin/env python3
"""Static ISA histogram of the gfx950 kernels of libdsv.

    python tools/isa_hist.py [--out profiles/rNN/isa_hist.json] [-D...]

Compiles every kernel translation unit of schnorr_amd/csrc/ (k_*.hip) to assembly (hipcc -S
--cuda-device-only, in parallel; no GPU needed), then
for every kernel reports registers, spills and the static count per opcode, and for the field
primitives (one-function probe kernels compiled from fe29.h / jubjub29.h) the split
"v_mad_u64_u32 vs everything else" that the roofline's MAD fraction is built on.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "schnorr_amd", "csrc")

PROBE = r'''
#include "%(csrc)s/fe29.h"
#include "%(csrc)s/jubjub29.h"
using namespace dsv;
// operands arrive in registers (loaded before s_barrier-free straight-line code); the histogram of
// a probe minus the histogram of k_probe_empty is the primitive itself
__global__ void k_probe_empty(Fe* p) { Fe a = p[threadIdx.x], b = p[threadIdx.x + 64]; p[threadIdx.x] = fe_add(a, b); }
__global__ void k_probe_fe_mul(Fe* p) { Fe a = p[threadIdx.x], b = p[threadIdx.x + 64]; p[threadIdx.x] = fe_mul(a, b); }
__global__ void k_probe_fe_sqr(Fe* p) { Fe a = p[threadIdx.x]; p[threadIdx.x] = fe_sqr(a); }
__global__ void k_probe_ext_double(Ext* p) { p[threadIdx.x] = ext_double(p[threadIdx.x]); }
__global__ void k_probe_ext_add_niels(Ext* p, Niels* n) { p[threadIdx.x] = ext_add_niels(p[threadIdx.x], n[threadIdx.x]); }
__global__ void k_probe_ext_add_aniels(Ext* p, ANiels* n) { p[threadIdx.x] = ext_add_aniels(p[threadIdx.x], n[threadIdx.x]); }
'''


def compile_asm(src, flags):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
           src] + flags
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def parse(path):
    text = open(path).read()
    kernels = {}
    # code: from "<name>:" to s_endpgm
    for m in re.finditer(r"^(_Z\w+|k_\w+):[^\n]*\n(.*?)\n\s+s_endpgm", text, re.S | re.M):
        c = collections.Counter()
        for line in m.group(2).split("\n"):
            t = line.strip().split()
            if t and re.match(r"^(v_|s_|global_|scratch_|buffer_|flat_|ds_)", t[0]):
                c[t[0]] += 1
        kernels[m.group(1)] = {"static": dict(c.most_common()), "static_total": sum(c.values())}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        k = kernels.setdefault(m.group(1), {})
        for key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count",
                    "private_segment_fixed_size"):
            mm = re.search(r"\.%s:\s+(\d+)" % key, m.group(2))
            if mm:
                k[key] = int(mm.group(1))
    return kernels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("flags", nargs="*", help="extra hipcc flags, e.g. -DDSV_FIXED_BITS=11")
    a = ap.parse_args()
    import concurrent.futures
    units = sorted(f for f in os.listdir(CSRC) if f.startswith("k_") and f.endswith(".hip"))
    lib = {}
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(units)) as ex:
        for part in ex.map(lambda u: parse(compile_asm(os.path.join(CSRC, u), a.flags)), units):
            lib.update(part)
    with tempfile.NamedTemporaryFile("w", suffix=".hip", delete=False) as f:
        f.write(PROBE % {"csrc": CSRC})
    probes = parse(compile_asm(f.name, a.flags))
    base = collections.Counter(next(v for k, v in probes.items() if "probe_empty" in k)["static"])
    prim = {}
    for name, v in probes.items():
        if "probe_empty" in name:
            continue
        short = re.search(r"k_probe_(\w+?)P", name).group(1)
        c = collections.Counter(v["static"])
        valu = {k: n for k, n in c.items() if k.startswith("v_")}
        mad = valu.get("v_mad_u64_u32", 0)
        # the probes' own address arithmetic / copies: what k_probe_empty also contains
        other = sum(valu.values()) - mad
        prim[short] = {"v_mad_u64_u32": mad, "other_valu": other, "valu_total": mad + other,
                       "mad_share": mad / max(1, mad + other),
                       "other_breakdown": {k: n for k, n in sorted(valu.items(), key=lambda x: -x[1])
                                           if k != "v_mad_u64_u32"}}
    rep = {"primitives": prim, "kernels": {}}
    for name, v in sorted(lib.items()):
        st = v.get("static", {})
        valu = sum(n for k, n in st.items() if k.startswith("v_"))
        rep["kernels"][name] = {
            "vgpr": v.get("vgpr_count"), "sgpr": v.get("sgpr_count"),
            "vgpr_spill": v.get("vgpr_spill_count"), "sgpr_spill": v.get("sgpr_spill_count"),
            "scratch_bytes": v.get("private_segment_fixed_size"),
            "static_instructions": v.get("static_total"), "static_valu": valu,
            "static_mad_u64_u32": st.get("v_mad_u64_u32", 0),
            "static_scratch_ops": sum(n for k, n in st.items() if k.startswith("scratch_")),
            "static_s_nop": st.get("s_nop", 0),
            "top": dict(list(st.items())[:12])}
    txt = json.dumps(rep, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    for k, v in prim.items():
        print("%-16s mad %4d  other %4d  share %.3f" % (k, v["v_mad_u64_u32"], v["other_valu"], v["mad_share"]))
    for k, v in rep["kernels"].items():
        print("%-70s vgpr %3s spill %4s sgpr_spill %3s static %6s nop %4s scratch_ops %s" % (
            k[:70], v["vgpr"], v["vgpr_spill"], v["sgpr_spill"], v["static_instructions"], v["static_s_nop"],
            v["static_scratch_ops"]))


if __name__ == "__main__":
    sys.exit(main())
