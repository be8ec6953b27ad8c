#!/usr/bin/env python3
"""Host-buffer entry point (dsv_verify_single on pageable numpy arrays) at 2^20 items:
verifies/s for the DSV_HOST_THREADS value of this process.  Run once per value."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
b = W.gen_single(n, seed=2321)
h = {k: b[k].cpu().numpy() for k in ("u", "R", "PK", "m")}
want = b["expected"].cpu().numpy()
E.verify_single(h["u"], h["R"], h["PK"], h["m"])
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ok = E.verify_single(h["u"], h["R"], h["PK"], h["m"])
    best = min(best, time.perf_counter() - t0)
assert (ok == want).all()
print("DSV_HOST_THREADS=%s n=2^%d: %.2f ms -> %.2f M verifies/s" % (
    os.environ.get("DSV_HOST_THREADS", "default"), n.bit_length() - 1, best * 1e3, n / best / 1e6))
