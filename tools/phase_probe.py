#!/usr/bin/env python3
"""Does it matter WHICH kernels share a SIMD?  2^20 single signatures as 16 sub-batches of 2^16 items on two
streams (each sub-batch = k_challenge then k_verify_fixed_half, one wave per SIMD each), in two arrangements:

  in phase    both streams start together: hash next to hash, verify next to verify
  anti phase  stream B first gets one extra verify-only launch, so from then on a hash on one stream runs
              next to a verify on the other (what the host pipeline's one-sub-batch chunks settle into:
              profiles/r05/host_timeline_e2e.txt)

Prints the time of the 16 sub-batches in both (the extra launch's own time measured separately and
subtracted).  Streams on different priority levels, so they never share a hardware queue.
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
n, part = 1 << 20, 1 << 16
b = W.gen_single(n, seed=2321)
want = b["expected"]
ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
ws = [torch.empty(E.workspace_bytes(part), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
c = torch.empty((part, 32), dtype=torch.uint8, device="cuda:0")
valid = torch.ones(part, dtype=torch.uint8, device="cuda:0")
okx = torch.zeros(part, dtype=torch.uint8, device="cuda:0")
wsx = torch.empty(E.workspace_bytes(part), dtype=torch.uint8, device="cuda:0")
streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
sl = lambda t, p: t[p * part:(p + 1) * part]


def parts(shift):
    if shift:  # one verify-only launch in front of stream B's sequence
        E.verify_core_dev(sl(b["u"], 0), c, valid, sl(b["PK"], 0), sl(b["R"], 0), okx, wsx, stream=streams[1])
    for p in range(n // part):
        k = p & 1
        E.verify_single_dev(sl(b["u"], p), sl(b["R"], p), sl(b["PK"], p), sl(b["m"], p), sl(ok, p), ws[k], stream=streams[k])


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


# the challenge of part 0 for the verify-only launch
from schnorr_amd import _lib  # noqa: E402
import ctypes  # noqa: E402
_lib.check(_lib.load().dsv_challenge_single_dev(ctypes.c_void_p(sl(b["R"], 0).data_ptr()), ctypes.c_void_p(sl(b["m"], 0).data_ptr()),
                                                ctypes.c_size_t(part), ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(valid.data_ptr()),
                                                ctypes.c_void_p(0)))
torch.cuda.synchronize()
extra = timed(lambda: E.verify_core_dev(sl(b["u"], 0), c, valid, sl(b["PK"], 0), sl(b["R"], 0), okx, wsx, stream=streams[1]))
for rnd in range(2):
    t_in = timed(lambda: parts(False))
    assert bool((ok == want).all())
    t_anti = timed(lambda: parts(True))
    assert bool((ok == want).all())
    print("in phase %.2f ms | anti phase %.2f ms incl. the extra verify-only launch (%.2f ms alone: it runs next to stream A's "
          "first sub-batch, so between %.2f and %.2f ms of it are extra) -> anti / in = %.3f .. %.3f" % (
              t_in, t_anti, extra, 0.0, extra, (t_anti - extra) / t_in, t_anti / t_in), flush=True)
