#!/usr/bin/env python3
"""verify_batch from typed objects (tools/libvb_e2e.so over include/dusk_schnorr.hpp): one-shot calls
and streamed calls (two batches in flight) beside the device-resident rate of the same scheme on the
same box, for the pipeline settings of THIS process (DSV_PIPE_PREP_STREAM, DSV_PIPE_FIRST_LOG2,
DSV_PIPE_CHUNK_LOG2, DSV_HOST_THREADS).  Run once per setting; one line per scheme.

    SCHEMES=single,double,vargen  LOG2N=20  VARGEN_LOG2N=18  python tools/e2e_probe.py
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
L = ctypes.CDLL(os.path.join(ROOT, "tools", "libvb_e2e.so"))
p = lambda a: ctypes.c_void_p(a.ctypes.data)
schemes = os.environ.get("SCHEMES", "single,double,vargen").split(",")
log2n = int(os.environ.get("LOG2N", "20"))
vlog2n = int(os.environ.get("VARGEN_LOG2N", "18"))
calls = int(os.environ.get("CALLS", "8"))
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items())
               if k.startswith("DSV_") and k != "DSV_PIPE_TRACE") or "defaults"


def dev_rate(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def proj_dev(pt, seed):
    """(u z, v z, z) on the device through the engine's own multiplier"""
    n = pt.shape[0]
    g = torch.Generator(device="cuda:0")
    g.manual_seed(seed)
    z = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda:0", generator=g)
    z[:, 31] = 0
    z[:, 0] |= 1
    zh = z.cpu().numpy()
    mul = lambda a: torch.from_numpy(E.debug_fq_mul(np.ascontiguousarray(a.cpu().numpy()), zh)).to("cuda:0")
    return torch.cat([mul(pt[:, :32]), mul(pt[:, 32:]), z], dim=1).contiguous()


for scheme in schemes:
    n = 1 << (vlog2n if scheme == "vargen" else log2n)
    if scheme == "single":
        b = W.gen_single(n, seed=2321)
        keys, kind = ("u", "R", "PK", "m"), 0
    elif scheme == "double":
        b = W.gen_double(n, seed=2322)
        keys, kind = ("u", "R", "Rp", "PK", "PKp", "m"), 1
    else:
        b = W.gen_vargen(n, seed=777)
        keys, kind = ("u", "R", "PK", "Gen", "m"), 2
    want = b["expected"].cpu().numpy()
    # device-resident reference of the same scheme: affine input, and projective input (what the
    # typed objects hold) for the single scheme
    okd = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(E.ext_workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    if kind == 0:
        dev = dev_rate(lambda: E.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], okd, ws))
        R3, PK3 = proj_dev(b["R"], 1), proj_dev(b["PK"], 2)
        dev_ext = dev_rate(lambda: E.verify_single_ext_dev(b["u"], R3, PK3, b["m"], okd, ws))
        del R3, PK3
    elif kind == 1:
        dev = dev_rate(lambda: E.verify_double_dev(b["u"], b["R"], b["Rp"], b["PK"], b["PKp"], b["m"], okd, ws))
        dev_ext = None
    else:
        dev = dev_rate(lambda: E.verify_vargen_dev(b["u"], b["R"], b["PK"], b["Gen"], b["m"], okd, ws))
        dev_ext = None
    assert (okd.cpu().numpy() == want).all()
    del ws
    h = [b[k].cpu().numpy() for k in keys]
    prep = (L.vb_e2e_prepare, L.vb_e2e_prepare_double, L.vb_e2e_prepare_vargen)[kind]
    run = (L.vb_e2e_run, L.vb_e2e_run_double, L.vb_e2e_run_vargen)[kind]
    bad = prep(*([p(a) for a in h] + [ctypes.c_size_t(n), ctypes.c_int(8)]))
    ok = np.zeros(n, dtype=np.uint8)
    ms = ctypes.c_double(0)
    one = []
    for rep in range(int(os.environ.get("REPS", "6"))):
        assert run(p(ok), ctypes.byref(ms)) == 0
        if rep:
            one.append(ms.value)
    assert (ok == want).all(), "one-shot verdicts differ"
    one.sort()
    res = {}
    for fl in ((1, 2) if os.environ.get("STREAMED", "1") == "1" else ()):
        best = 1e9
        for rep in range(3):
            ok[:] = 7
            rc = L.vb_e2e_run_streamed(ctypes.c_int(kind), ctypes.c_int(calls), ctypes.c_int(fl), p(ok), ctypes.byref(ms))
            assert rc == 0, "streamed run: rc %d" % rc
            assert (ok == want).all(), "streamed verdicts differ"
            best = min(best, ms.value / calls)
        res[fl] = best
    L.vb_e2e_release()
    line = "[%s] %-6s n=2^%d bad=%d: device-resident %.2f ms = %.1f M/s" % (tag, scheme, n.bit_length() - 1, bad, dev * 1e3, n / dev / 1e6)
    if dev_ext:
        line += " (projective %.2f ms = %.1f M/s)" % (dev_ext * 1e3, n / dev_ext / 1e6)
    ref = dev_ext or dev
    line += " | one-shot best %.2f median %.2f ms = %.1f M/s (%.3f x)" % (one[0], one[len(one) // 2], n / one[0] / 1e3, ref * 1e3 / one[0])
    if res:
        line += " | back-to-back %.2f ms = %.1f M/s (%.3f x) | two in flight %.2f ms = %.1f M/s (%.3f x)" % (
            res[1], n / res[1] / 1e3, ref * 1e3 / res[1], res[2], n / res[2] / 1e3, ref * 1e3 / res[2])
    print(line, flush=True)
    del b, okd
