#!/usr/bin/env python3
"""Which kernels may share the SIMDs?  LOCKED pairings of the two kernels of a verification on two streams
(every pair starts together: each stream waits for the other's previous kernel), 2^16-item sub-batches:

  hash || hash, then verify || verify      what the device-resident entry points settle into
  hash || verify, then verify || hash      what the host pipeline's one-sub-batch chunks can lock into
                                           (profiles/r05/host_timeline_e2e.txt: both kernels 1.35 ms)

Prints the time per pair of sub-batches for both, and the kernels' times alone (one stream)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from schnorr_amd import _lib  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
part, pairs = 1 << 16, 8
n = part * 2 * pairs
b = W.gen_single(n, seed=2321)
want = b["expected"]
ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
c = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
valid = torch.empty(n, dtype=torch.uint8, device="cuda:0")
ws = [torch.empty(E.workspace_bytes(part), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
S = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
L = _lib.load()
sl = lambda t, p: t[p * part:(p + 1) * part]
P = lambda t: ctypes.c_void_p(t.data_ptr())


def hash_(p, k):
    _lib.check(L.dsv_challenge_single_dev(P(sl(b["R"], p)), P(sl(b["m"], p)), ctypes.c_size_t(part), P(sl(c, p)), P(sl(valid, p)),
                                          ctypes.c_void_p(S[k].cuda_stream)))


def verify(p, k):
    E.verify_core_dev(sl(b["u"], p), sl(c, p), sl(valid, p), sl(b["PK"], p), sl(b["R"], p), sl(ok, p), ws[k], stream=S[k])


def barrier():
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    for k in range(2):
        ev[k].record(S[k])
    for k in range(2):
        S[k].wait_event(ev[1 - k])


def in_phase():
    for i in range(pairs):
        hash_(2 * i, 0), hash_(2 * i + 1, 1)
        barrier()
        verify(2 * i, 0), verify(2 * i + 1, 1)
        barrier()


def anti_phase():
    hash_(0, 0)                       # stream A is one kernel ahead
    barrier()
    for i in range(pairs):
        verify(2 * i, 0), hash_(2 * i + 1, 1)
        barrier()
        if i + 1 < pairs:
            hash_(2 * i + 2, 0)
        verify(2 * i + 1, 1)
        barrier()


def alone(fn):
    def run():
        for p in range(2 * pairs):
            fn(p, 0)
    return run


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


t_h = timed(alone(hash_)) / (2 * pairs)
t_v = timed(alone(verify)) / (2 * pairs)
for rnd in range(2):
    ti = timed(in_phase)
    assert bool((ok == want).all())
    ta = timed(anti_phase)
    assert bool((ok == want).all())
    print("alone on one stream: hash %.3f ms, verify %.3f ms per sub-batch | locked in phase %.3f ms per pair of sub-batches | "
          "locked anti phase %.3f ms per pair (x%.3f)" % (t_h, t_v, ti / pairs, ta / pairs, ta / ti), flush=True)
