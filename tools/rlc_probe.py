#!/usr/bin/env python3
"""Time dsv_verify_<scheme>_rlc_dev beside dsv_verify_<scheme>_dev (device-resident inputs):

    python tools/rlc_probe.py [log2n=20] [window_bits=0] [reps=10] [scheme=single|double|vargen]

Workloads (verdicts checked against the construction-time pattern every time):
  all valid          steady state of a caller whose batches are valid: one aggregate per group decides
  all valid, guarded the same for a caller whose batches fail now and then: a second stage is enqueued and switched off
  all valid, split   the same while the device's history says "batches fail": sub-groups + sample
  one wrong, unguarded  ONE wrong signature for a caller that never saw one: whole-group fallback
  one wrong, guarded    ... the first one after a run of valid batches: the second stage localises it
  one wrong, split   ONE wrong signature while the history says so: only its sub-group takes the fallback
  1/16 tampered      the graded workload (wrong items throughout): the sample skips the aggregates
The calls are enqueue-only (accepted comes back through a pinned word); the timing loop synchronises
once per call, like the per-signature baseline."""
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
acc_word = torch.zeros(1, dtype=torch.int32).pin_memory()


def timed(fn, before=None):
    if before:
        before()
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if before:
            before()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


good = gen(n, seed=2321, tamper=False)
graded = gen(n, seed=2321, tamper=True)
one = {k: v.clone() for k, v in good.items()}
victim = (5 * n) // 8 + 77          # inside the third of four sub-groups, far from the sample's reach on average
one["u"][victim, 3] ^= 0x10
one["expected"][victim] = 0
big = n >= (1 << 17 if scheme == "single" else 1 << 14) or bits != 0     # (automatic bits: smaller groups skip the aggregate)

best0, med0 = timed(lambda: plain(*[good[k] for k in COLS], ok, ws))
assert torch.equal(ok, good["expected"])
print("%s n=2^%d bits=%d per-signature path          %.3f ms (median %.3f) = %.1f M/s" % (
    scheme, n.bit_length() - 1, bits, best0, med0, n / best0 / 1e3), flush=True)
# (history, long history): (0, 0) steady state of a caller whose batches never fail; (0, 128) "guarded": batches
# fail now and then; (8, 128) "split": one failed within the last eight calls
for label, b, history, expect in (("all valid", good, (0, 0), big), ("all valid, guarded", good, (0, 128), big),
                                  ("all valid, split", good, (8, 128), big),
                                  ("one wrong, unguarded", one, (0, 0), False), ("one wrong, guarded", one, (0, 128), False),
                                  ("one wrong, split", one, (8, 128), False),
                                  ("1/16 tampered", graded, (8, 128), False)):
    cols = [b[k] for k in COLS]
    accs = []

    def call():
        rlc(*cols, ok, ws, window_bits=bits, accepted_out=acc_word)

    def check():
        torch.cuda.synchronize()
        accs.append(int(acc_word[0]))

    ok.zero_()
    ts = []
    for _ in range(reps + 1):
        E.rlc_history(0, history[0])
        E.rlc_history_long(0, history[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        check()
        assert torch.equal(ok, b["expected"]), label
    ts = sorted(ts[1:])
    assert all(a == int(expect) for a in accs), (label, accs)
    best, med = ts[0], ts[len(ts) // 2]
    print("%s n=2^%d bits=%d %-21s rlc %.3f ms (median %.3f) = %.1f M/s | x%.2f the per-signature path" % (
        scheme, n.bit_length() - 1, bits, label, best, med, n / best / 1e3, best0 / best), flush=True)
