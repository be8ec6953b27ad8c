#!/usr/bin/env python3
"""Host-buffer entry points on pageable numpy arrays at 2^LOG2N items (default 2^20): verifies/s of
dsv_verify_single (affine points, 192 B per item), dsv_verify_single_ext (projective points, 256 B)
and dsv_verify_single_wire (serialized records, 128 B) for the DSV_HOST_THREADS / DSV_PIPE_CHUNK_LOG2
values of this process.  Run once per setting."""
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
z = np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8)
z[:, 31] = 0
z[:, 0] |= 1
proj = lambda a: np.concatenate([E.debug_fq_mul(np.ascontiguousarray(a[:, :32]), z),
                                 E.debug_fq_mul(np.ascontiguousarray(a[:, 32:]), z), z], axis=1)
R3, PK3 = proj(h["R"]), proj(h["PK"])
sig = np.ascontiguousarray(np.concatenate([h["u"], E.compress_points(h["R"])], axis=1))
pk = E.compress_points(h["PK"])


def best_of(fn, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        ok = fn()
        best = min(best, time.perf_counter() - t0)
    assert (ok == want).all()
    return best


def dev_rate(fn, reps=10):
    import torch
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if os.environ.get("WITH_DEVICE", "1") == "1":
    import torch
    dv = lambda a: torch.from_numpy(a).to("cuda:0")
    okd = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(max(E.ext_workspace_bytes(n), E.wire_workspace_bytes(n)), dtype=torch.uint8, device="cuda:0")
    dR3, dPK3, dsig, dpk = dv(R3), dv(PK3), dv(sig), dv(pk)
    dres = {"affine": dev_rate(lambda: E.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], okd, ws)),
            "ext": dev_rate(lambda: E.verify_single_ext_dev(b["u"], dR3, dPK3, b["m"], okd, ws)),
            "wire": dev_rate(lambda: E.verify_single_wire_dev(dsig, dpk, b["m"], okd, ws))}
    assert (okd.cpu().numpy() == want).all()
    print("device-resident, same box: " + "  ".join("%s %.2f ms = %.2f M/s" % (k, v * 1e3, n / v / 1e6) for k, v in dres.items()))
    del dR3, dPK3, dsig, dpk, ws

res = {"affine": best_of(lambda: E.verify_single(h["u"], h["R"], h["PK"], h["m"])),
       "ext": best_of(lambda: E.verify_single_ext(h["u"], R3, PK3, h["m"])),
       "wire": best_of(lambda: E.verify_single_wire(sig, pk, h["m"]))}
print("DSV_HOST_THREADS=%s DSV_PIPE_CHUNK_LOG2=%s n=2^%d: " % (
    os.environ.get("DSV_HOST_THREADS", "default"), os.environ.get("DSV_PIPE_CHUNK_LOG2", "default"),
    n.bit_length() - 1) + "  ".join("%s %.2f ms = %.2f M/s" % (k, v * 1e3, n / v / 1e6) for k, v in res.items()))
if os.environ.get("WITH_DEVICE", "1") == "1":
    print("host / device-resident: " + "  ".join("%s %.3f" % (k, dres[k] / res[k]) for k in res))
