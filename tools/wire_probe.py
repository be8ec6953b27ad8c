#!/usr/bin/env python3
"""Throughput of the serialized-record entry point dsv_verify_single_wire (host buffers:
64-byte Signature + 32-byte PublicKey + 32-byte message per item) at 2^20 items."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401

from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
b = W.gen_single(n, seed=2321)
h = {k: b[k].cpu().numpy() for k in ("u", "R", "PK", "m")}
want = b["expected"].cpu().numpy()
sig = np.concatenate([h["u"], E.compress_points(h["R"])], axis=1)
pk = E.compress_points(h["PK"])
E.verify_single_wire(sig, pk, h["m"])
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    ok = E.verify_single_wire(sig, pk, h["m"])
    best = min(best, time.perf_counter() - t0)
assert (ok == want).all()
print("verify_single_wire n=2^%d: %.2f ms -> %.2f M verifies/s (128 B/item over PCIe)" % (
    n.bit_length() - 1, best * 1e3, n / best / 1e6))
