#!/usr/bin/env python3
"""How much does a SMALL kernel cost a pair of kernels that exactly fill the chip?
The device-resident path runs 2^16-item sub-batches on two streams: 1024 + 1024 waves on 1024 SIMDs with two
wave slots each — an exact fit.  Beside it, on a third (high-priority) stream: a train of small launches
(k_fixed_base_points on 8192 items = 128 waves of ~0.15 ms, the footprint of the host pipeline's
normalisation kernel).  Prints the main call alone, the train alone, both together, and the cost of the train
in units of its own work — 1.0 would be "it costs what it computes"."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
n = 1 << 20
part = int(os.environ.get("PART", str(1 << 16)))
small = int(os.environ.get("SMALL", "8192"))
b = W.gen_single(n, seed=2321)
want = b["expected"]
ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
ws = [torch.empty(E.workspace_bytes(1 << 16), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
sk = torch.randint(0, 256, (small, 32), dtype=torch.uint8, device="cuda:0")
sk[:, 31] = 0
pk = torch.empty((small, 64), dtype=torch.uint8, device="cuda:0")
lanes = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
side = torch.cuda.Stream(priority=-1)
sl = lambda t, lo, hi: t[lo:hi]


def main_call():
    p = 0
    for lo in range(0, n, part):
        hi = min(n, lo + part)
        k = p & 1
        E.verify_single_dev(sl(b["u"], lo, hi), sl(b["R"], lo, hi), sl(b["PK"], lo, hi), sl(b["m"], lo, hi), sl(ok, lo, hi), ws[k], stream=lanes[k])
        p += 1


def train(count):
    for _ in range(count):
        E.public_keys_dev(sk, 0, pk, stream=side)


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


for count in (0, 20, 40, 80):
    t_main = timed(main_call)
    assert bool((ok == want).all())
    t_train = timed(lambda: train(count)) if count else 0.0
    t_both = timed(lambda: (train(count), main_call())) if count else t_main
    work = count * small / n * 1.26          # k_fixed_base_points: 1.26 ms per 2^20 at full occupancy
    print("part %d, train of %d x %d items (work %.2f ms at full occupancy; alone %.2f ms): main alone %.2f ms, together %.2f ms -> +%.2f ms = x%.1f its work" % (
        part, count, small, work, t_train, t_main, t_both, t_both - t_main, (t_both - t_main) / work if work else 0), flush=True)
