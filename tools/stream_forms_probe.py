#!/usr/bin/env python3
"""Two host calls in flight (two threads, K calls each) for every input form of the single scheme — affine
bytes (no preprocessing), projective bytes (normalisation), Montgomery limbs (normalisation + scalar
reductions), wire records (decompression per sub-batch) — beside the device-resident rate of the same
form: which part of the gap between the streamed host path and HBM-resident input is the preprocessing?"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch  # noqa: E402

import oracle_lib as O  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
n = 1 << 20
K = int(os.environ.get("CALLS", "6"))
b = W.gen_single(n, seed=2321)
want = b["expected"].cpu().numpy()
h = {k: b[k].cpu().numpy() for k in ("u", "R", "PK", "m")}
z = np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8)
z[:, 31] = 0
z[:, 0] |= 1
proj = lambda a: np.concatenate([E.debug_fq_mul(np.ascontiguousarray(a[:, :32]), z),
                                 E.debug_fq_mul(np.ascontiguousarray(a[:, 32:]), z), z], axis=1)
R3, PK3 = proj(h["R"]), proj(h["PK"])
mont = [O.to_mont(h["u"], fr=True), O.to_mont(R3), O.to_mont(PK3), O.to_mont(h["m"])]
sig = np.ascontiguousarray(np.concatenate([h["u"], E.compress_points(h["R"])], axis=1))
pk = E.compress_points(h["PK"])
dv = lambda a: torch.from_numpy(a).to("cuda:0")
okd = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
ws = torch.empty(max(E.mont_workspace_bytes(n), E.wire_workspace_bytes(n)), dtype=torch.uint8, device="cuda:0")


def dev_rate(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    assert (okd.cpu().numpy() == want).all()
    return (time.perf_counter() - t0) / reps * 1e3


dR3, dPK3, dmont, dsig, dpk = dv(R3), dv(PK3), [dv(a) for a in mont], dv(sig), dv(pk)
forms = {
    "affine": (lambda: E.verify_single(h["u"], h["R"], h["PK"], h["m"]),
               lambda: E.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], okd, ws)),
    "projective": (lambda: E.verify_single_ext(h["u"], R3, PK3, h["m"]),
                   lambda: E.verify_single_ext_dev(b["u"], dR3, dPK3, b["m"], okd, ws)),
    "limbs": (lambda: E.verify_single_mont(*mont), lambda: E.verify_single_mont_dev(*dmont, okd, ws)),
    "wire": (lambda: E.verify_single_wire(sig, pk, h["m"]), lambda: E.verify_single_wire_dev(dsig, dpk, b["m"], okd, ws)),
}
for name, (host, dev) in forms.items():
    d = dev_rate(dev)
    host()
    one = []
    for _ in range(5):
        t0 = time.perf_counter()
        got = host()
        one.append((time.perf_counter() - t0) * 1e3)
    assert (got == want).all()
    errs = []

    def worker():
        for _ in range(K):
            if not (host() == want).all():
                errs.append(1)

    best = 1e9
    for _ in range(3):
        th = [threading.Thread(target=worker) for _ in range(2)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        best = min(best, (time.perf_counter() - t0) * 1e3 / (2 * K))
    assert not errs
    print("%-10s device-resident %.2f ms | one call %.2f ms (x%.3f) | two threads, %d calls each: %.2f ms per call (x%.3f)" % (
        name, d, min(one), d / min(one), K, best, d / best), flush=True)
