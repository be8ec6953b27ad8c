#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak: many seeds x (single, double, var-generator, projective-input,
wire) batches with the harness's tamper classes, verdict vectors compared bit for bit.  Prints one line per seed so the
run shows progress; exits non-zero on the first difference.

    python tools/soak.py [--seeds N] [--items M]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import harness as H  # noqa: E402
import oracle_lib as O  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--items", type=int, default=8192)
    ap.add_argument("--first-seed", type=int, default=1000)
    args = ap.parse_args()
    E.init(0)
    threads = min(16, len(os.sched_getaffinity(0)))
    total = 0
    t0 = time.time()
    for seed in range(args.first_seed, args.first_seed + args.seeds):
        n = args.items
        s = O.keygen_sign_single(n, seed, nthreads=threads)
        H.tamper(s, period=3 + seed % 11)
        want = O.verify_single(s["u"], s["R"], s["PK"], s["m"], nthreads=threads)
        got = E.verify_single(s["u"], s["R"], s["PK"], s["m"])
        assert np.array_equal(got, want), ("single", seed, np.nonzero(got != want)[0][:8])
        d = O.keygen_sign_double(n // 2, seed + 7, nthreads=threads)
        H.tamper(d, kind_single=False, period=2 + seed % 7)
        want = O.verify_double(d["u"], d["R"], d["Rp"], d["PK"], d["PKp"], d["m"], nthreads=threads)
        got = E.verify_double(d["u"], d["R"], d["Rp"], d["PK"], d["PKp"], d["m"])
        assert np.array_equal(got, want), ("double", seed, np.nonzero(got != want)[0][:8])
        v = O.keygen_sign_vargen(n // 2, seed + 13, nthreads=threads)
        H.tamper(v, period=2 + seed % 5)
        want = O.verify_vargen(v["u"], v["R"], v["PK"], v["Gen"], v["m"], nthreads=threads)
        got = E.verify_vargen(v["u"], v["R"], v["PK"], v["Gen"], v["m"])
        assert np.array_equal(got, want), ("vargen", seed, np.nonzero(got != want)[0][:8])
        # the same var-generator and double batches as projective points (random z per point): the
        # device's to_hash_inputs must not change a verdict (equal points verify alike whatever
        # their z: oracle_verify_*_ext is checked against that in tests/test_gpu_r03.py)
        rng = np.random.default_rng(seed)

        def proj(a):
            z = rng.integers(0, 256, (a.shape[0], 32), dtype=np.uint8)
            z[:, 31] = 0
            z[:, 0] |= 1
            return np.concatenate([E.debug_fq_mul(np.ascontiguousarray(a[:, :32]), z),
                                   E.debug_fq_mul(np.ascontiguousarray(a[:, 32:]), z), z], axis=1)

        got = E.verify_vargen_ext(v["u"], proj(v["R"]), proj(v["PK"]), proj(v["Gen"]), v["m"])
        # (a tampered coordinate >= q stays >= q only in the affine form: compare where both are canonical)
        canon = np.ones(len(want), bool)
        for k in ("R", "PK", "Gen"):
            canon &= (v[k][:, 31] < 0x73) & (v[k][:, 63] < 0x73)
        assert np.array_equal(got[canon], want[canon]), ("vargen ext", seed)
        wd = O.verify_double(d["u"], d["R"], d["Rp"], d["PK"], d["PKp"], d["m"], nthreads=threads)
        got = E.verify_double_ext(d["u"], proj(d["R"]), proj(d["Rp"]), proj(d["PK"]), proj(d["PKp"]), d["m"])
        canon = np.ones(len(wd), bool)
        for k in ("R", "Rp", "PK", "PKp"):
            canon &= (d[k][:, 31] < 0x73) & (d[k][:, 63] < 0x73)
        assert np.array_equal(got[canon], wd[canon]), ("double ext", seed)
        # wire formats of the same single batch
        sig = np.concatenate([s["u"], O.compress(s["R"])], axis=1)
        got = E.verify_single_wire(sig, O.compress(s["PK"]), s["m"])
        want = O.verify_single_wire(sig[:512], O.compress(s["PK"])[:512], s["m"][:512])
        assert np.array_equal(got[:512], want), ("wire", seed)
        total += 3 * n + n // 2
        print("seed %d ok  (%d verdicts compared, %.0f s)" % (seed, total, time.time() - t0), flush=True)
    print("SOAK OK: %d seeds, %d verdicts, all bit-exact" % (args.seeds, total))


if __name__ == "__main__":
    main()
