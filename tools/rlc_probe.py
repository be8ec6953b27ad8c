#!/usr/bin/env python3
"""Time dsv_verify_single_rlc_dev beside dsv_verify_single_dev (device-resident inputs):

    python tools/rlc_probe.py [log2n=20] [window_bits=0] [reps=10] [scheme=single|double|vargen]

all-valid batch (the aggregate decides) and the graded workload (1/16 tampered: aggregate fails, the
per-signature kernels decide).  Verdicts checked against the construction-time pattern."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
scheme = sys.argv[4] if len(sys.argv) > 4 else "single"
COLS = {"single": ("u", "R", "PK", "m"), "double": ("u", "R", "Rp", "PK", "PKp", "m"), "vargen": ("u", "R", "PK", "Gen", "m")}[scheme]
gen = getattr(W, "gen_" + scheme)
rlc = getattr(E, "verify_%s_rlc_dev" % scheme)
plain = getattr(E, "verify_%s_dev" % scheme)
ws = torch.empty(E.rlc_workspace_bytes(n, bits), dtype=torch.uint8, device="cuda:0")
ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


for label, tamper in (("all valid", False), ("1/16 tampered", True)):
    b = gen(n, seed=2321, tamper=tamper)
    cols = [b[k] for k in COLS]
    acc = []
    best, med = timed(lambda: acc.append(rlc(*cols, ok, ws, window_bits=bits)))
    expect = (not tamper) and (bits != 0 or n >= 1 << 17)   # (automatic bits: groups below 2^17 items skip the aggregate)
    assert torch.equal(ok, b["expected"]) and all(a == expect for a in acc)
    ok.zero_()
    best0, med0 = timed(lambda: plain(*cols, ok, ws))
    assert torch.equal(ok, b["expected"])
    print(scheme + " n=2^%d bits=%d %-14s rlc %.3f ms (median %.3f) = %.1f M/s | per-signature %.3f ms (median %.3f) = %.1f M/s | x%.2f" % (
        n.bit_length() - 1, bits, label, best, med, n / best / 1e3, best0, med0, n / best0 / 1e3, best0 / best), flush=True)
