#!/usr/bin/env python3
"""Randomised soak of the Montgomery-limb entry points against the oracle, bit for bit:
    python tools/soak_mont.py [seeds=60] [n=257]
Per seed and scheme (single, double, var-generator): a signed + tampered batch with a random z per
point and the planted encodings the Rust types cannot hold (tests/mont_cases.py), through the dense
host entry point, the strided columns of record arrays laid out like the Rust structs, and the
device-pointer form.  Prints one line per seed and a total."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import mont_cases as C  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(sys.argv[2]) if len(sys.argv) > 2 else 257
E.init(0)
total = bad = 0
t0 = time.time()
for seed in range(1000, 1000 + seeds):
    line = []
    for scheme in ("single", "double", "vargen"):
        cols, want = C.mont_case(scheme, n, seed * 3 + len(scheme), period=3 + seed % 11)
        got = [getattr(E, "verify_%s_mont" % scheme)(*cols),
               E.verify_mont_cols(scheme, C.as_records(scheme, cols)[3])]
        ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        ws = torch.empty(E.mont_workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
        getattr(E, "verify_%s_mont_dev" % scheme)(*[torch.from_numpy(c).to("cuda:0") for c in cols], ok, ws)
        torch.cuda.synchronize()
        got.append(ok.cpu().numpy())
        for g in got:
            total += n
            bad += int((g != want).sum())
        line.append("%s %d/%d valid" % (scheme, int(want.sum()), n))
    print("seed %d: %s  (%.0f s)" % (seed, ", ".join(line), time.time() - t0), flush=True)
print("soak_mont: %d verdicts of three schemes x three entry-point forms compared with the oracle, %d different"
      % (total, bad))
sys.exit(1 if bad else 0)
