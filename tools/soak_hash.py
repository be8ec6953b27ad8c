#!/usr/bin/env python3
"""Challenge-hash soak: GPU (matrix-core Hades, hades_mfma.h) against the oracle on random and on
structured field elements — byte patterns that push the signed digits of the matrix-core operands
to their extremes (0x00 / 0x7f / 0x80 / 0xff runs), values next to 0, q and 2^k.  Single and double
hash, bit for bit.  One line per seed; exits non-zero on the first difference.

    python tools/soak_hash.py [--seeds N] [--items M]
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import oracle_lib as O  # noqa: E402
import pymodel as M  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402


def felts(rng, n):
    raw = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    kind = rng.integers(0, 8, size=n)
    pat = np.array([0x00, 0x7F, 0x80, 0xFF], dtype=np.uint8)
    for i in range(n):
        k = kind[i]
        if k == 0:      # runs of extreme bytes
            a, b = sorted(rng.integers(0, 33, size=2))
            raw[i, a:b] = pat[rng.integers(0, 4)]
        elif k == 1:    # every byte extreme
            raw[i] = pat[rng.integers(0, 4, size=32)]
        elif k == 2:    # near a power of two / near q / near 0
            v = [0, 1, M.Q - 1, M.Q - 2, (1 << int(rng.integers(1, 255))) - int(rng.integers(0, 2))][int(rng.integers(0, 5))]
            raw[i] = np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)
    # canonical: below q
    out = np.empty_like(raw)
    for i in range(n):
        v = int.from_bytes(raw[i].tobytes(), "little") % M.Q
        out[i] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=10)
    ap.add_argument("--items", type=int, default=16384)
    ap.add_argument("--first-seed", type=int, default=5000)
    args = ap.parse_args()
    E.init(0)
    threads = min(16, len(os.sched_getaffinity(0)))
    t0 = time.time()
    total = 0
    for seed in range(args.first_seed, args.first_seed + args.seeds):
        rng = np.random.default_rng(seed)
        n = args.items - int(rng.integers(0, 300))          # ragged: partly empty waves / workgroups
        R = np.concatenate([felts(rng, n), felts(rng, n)], axis=1)
        Rp = np.concatenate([felts(rng, n), felts(rng, n)], axis=1)
        m = felts(rng, n)
        parts = np.array_split(np.arange(n), threads)
        with ThreadPoolExecutor(threads) as ex:
            ws = list(ex.map(lambda ix: O.challenge_single(R[ix], m[ix]), parts))
            wd = list(ex.map(lambda ix: O.challenge_double(R[ix], Rp[ix], m[ix]), parts))
        want_s, want_d = np.concatenate(ws), np.concatenate(wd)
        got_s, got_d = E.challenge_single(R, m), E.challenge_double(R, Rp, m)
        bad = int((got_s != want_s).any(axis=1).sum()) + int((got_d != want_d).any(axis=1).sum())
        total += 2 * n
        print("seed %d: n=%d  mismatches %d  (%.0f s, %d hashes so far)" % (seed, n, bad, time.time() - t0, total),
              flush=True)
        if bad:
            sys.exit(1)
    print("OK: %d hashes identical" % total)


if __name__ == "__main__":
    main()
