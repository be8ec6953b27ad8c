#!/usr/bin/env python3
"""Soak of the host pipeline with SEVERAL CALLS IN FLIGHT (r05), bit for bit against the oracle:

    python tools/soak_jobs.py [rounds=40] [seed=1]

Every round builds a few signed + tampered batches (three schemes, projective points with a random z,
Montgomery limbs, the planted encodings the Rust types cannot hold: tests/mont_cases.py) tiled to random
sizes from one small chunk to several pipeline chunks with ragged tails, then runs them ALL AT ONCE:
some as jobs (dsv_verify_*_mont_cols_submit, more than dsv_max_in_flight() of them), some as blocking
calls from threads of their own (affine bytes, limbs, wire records, the typed-object fast accept
dsv_verify_*_mont_cols_rlc), with 1 to 4 copy threads.  Whatever
shares the compute lanes, every call's verdicts must be the oracle's.  One line per round, a total, exit
code 1 on any difference.  The oracle (test infrastructure) only checks; nothing here is timed.
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import harness as H  # noqa: E402
import mont_cases as C  # noqa: E402
import oracle_lib as O  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
E.init(0)
SIZES = [300, 1 << 14, (1 << 15) + 77, (1 << 16) + 1, (1 << 16) + (1 << 15) + 333, (1 << 17) + 4099,
         (1 << 18) + (1 << 16) + 5, 3 * (1 << 17) + 17]
total = bad = 0
t0 = time.time()
for rd in range(rounds):
    calls = []   # (label, thunk returning verdicts or a job, expected)
    for k in range(int(rng.integers(3, 7))):
        scheme = ("single", "double", "vargen")[int(rng.integers(0, 3))]
        n = int(SIZES[int(rng.integers(0, len(SIZES)))])
        if scheme != "single":
            n = min(n, (1 << 17) + 4099)
        base = int(rng.integers(150, 400))
        cols, want = C.mont_case(scheme, base, int(rng.integers(1, 1 << 30)), period=int(rng.integers(3, 12)))
        reps = -(-n // base)
        tcols = [np.ascontiguousarray(np.tile(c, (reps, 1))[:n]) for c in cols]
        twant = np.tile(want, reps)[:n]
        form = int(rng.integers(0, 4))
        if form == 3:   # the typed-object fast accept (its verdicts must be the oracle's whether or not it accepts)
            views = C.as_records(scheme, tcols)[3]
            calls.append(("%s fast accept n=%d" % (scheme, n), ("call", lambda v=views, s=scheme: E.verify_mont_cols_rlc(s, v)[0], None), twant))
        elif form == 0:
            views = C.as_records(scheme, tcols)[3]
            calls.append(("%s job n=%d" % (scheme, n), ("job", scheme, views), twant))
        elif form == 1:
            calls.append(("%s limbs n=%d" % (scheme, n), ("call", getattr(E, "verify_%s_mont" % scheme), tcols), twant))
        else:
            views = C.as_records(scheme, tcols)[3]
            calls.append(("%s columns n=%d" % (scheme, n), ("call", lambda v=views, s=scheme: E.verify_mont_cols(s, v), None), twant))
    # one affine-bytes and one wire call of the single scheme beside them
    d = O.keygen_sign_single(256, int(rng.integers(1, 1 << 30)), nthreads=4)
    H.tamper(d)
    w = O.verify_single(d["u"], d["R"], d["PK"], d["m"], nthreads=4)
    n = int(SIZES[int(rng.integers(0, len(SIZES)))])
    reps = -(-n // 256)
    t = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    u, R, PK, m = t(d["u"]), t(d["R"]), t(d["PK"]), t(d["m"])
    tw = np.tile(w, reps)[:n]
    calls.append(("single affine n=%d" % n, ("call", lambda: E.verify_single(u, R, PK, m), None), tw))
    sig = np.ascontiguousarray(np.concatenate([u, E.compress_points(R)], axis=1))
    pk = E.compress_points(PK)
    calls.append(("single wire n=%d" % n, ("call", lambda: E.verify_single_wire(sig, pk, m), None), tw))
    E.set_host_threads(int(rng.integers(1, 5)))
    results = [None] * len(calls)
    errors = []

    def run_call(i, fn, args):
        try:
            results[i] = fn(*args) if args is not None else fn()
        except Exception as e:  # noqa: BLE001
            errors.append("%s: %r" % (calls[i][0], e))

    threads, jobs = [], []
    order = rng.permutation(len(calls))
    for i in order:
        kind = calls[i][1]
        if kind[0] == "job":
            jobs.append((i, E.submit_mont_cols(kind[1], kind[2])))
        else:
            th = threading.Thread(target=run_call, args=(i, kind[1], kind[2]))
            th.start()
            threads.append(th)
    for i, j in jobs:
        results[i] = j.wait()
    for th in threads:
        th.join()
    E.set_host_threads(0)
    diffs = []
    for (label, _, want), got in zip(calls, results):
        total += len(want)
        if got is None or not np.array_equal(got, want):
            nb = len(want) if got is None else int((got != want).sum())
            bad += nb
            diffs.append("%s: %d different" % (label, nb))
    print("round %d: %d calls at once (%d jobs, %d fast accepts), %d verdicts%s  (%.0f s)" % (
        rd, len(calls), len(jobs), sum("fast accept" in c[0] for c in calls), sum(len(c[2]) for c in calls),
        ("  DIFFERENT: " + "; ".join(diffs + errors)) if diffs or errors else "", time.time() - t0), flush=True)
    if errors:
        bad += 1
print("soak_jobs: %d verdicts from calls in flight together compared with the oracle, %d different" % (total, bad))
sys.exit(1 if bad else 0)
