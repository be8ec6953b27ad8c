#!/usr/bin/env python3
"""Regenerate tests/golden/predicted_reference.json: what THIS repository predicts the real
dusk-schnorr 0.18 prints for the reference's own test seeds (tests/schnorr.rs:16 seed 2321,
benches/signature.rs:81 seed 0xbeef): draw order sk, message, then (inside sign) the nonce, each
64 keystream bytes through from_bytes_wide.

UNVERIFIED PREDICTION (parity unpinned): it chains the restated StdRng (tests/refrng.py), the
oracle's field reduction, group law, hash and serialisation.  rust/dusk-schnorr-gpu/src/bin/
golden_gen.rs prints the same fields from the real crate; byte equality pins all of them at once.
"""
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
import refrng  # noqa: E402


def run(seed, count):
    rng = refrng.StdRng(seed)
    recs = []
    for i in range(count):
        skw = np.frombuffer(rng.fill_bytes(64), np.uint8).reshape(1, 64).copy()
        mw = np.frombuffer(rng.fill_bytes(64), np.uint8).reshape(1, 64).copy()
        rw = np.frombuffer(rng.fill_bytes(64), np.uint8).reshape(1, 64).copy()
        out = {k: np.zeros((1, s), np.uint8) for k, s in
               (("sk", 32), ("m", 32), ("u", 32), ("R", 64), ("PK", 64))}
        O.lib().oracle_keygen_sign_single(O._p(skw), O._p(mw), O._p(rw), ctypes.c_size_t(1),
                                          O._p(out["sk"]), O._p(out["m"]), O._p(out["u"]),
                                          O._p(out["R"]), O._p(out["PK"]), ctypes.c_int(1))
        sig_bytes = bytes(out["u"][0]) + bytes(O.compress(out["R"])[0])
        pk_bytes = bytes(O.compress(out["PK"])[0])
        ok = int(O.verify_single(out["u"], out["R"], out["PK"], out["m"])[0])
        recs.append({"i": i, "sk": bytes(out["sk"][0]).hex(), "m": bytes(out["m"][0]).hex(),
                     "u": bytes(out["u"][0]).hex(), "R": bytes(out["R"][0]).hex(),
                     "PK": bytes(out["PK"][0]).hex(), "sig_bytes": sig_bytes.hex(),
                     "pk_bytes": pk_bytes.hex(), "verdict": ok})
    return recs


def main():
    out = {"note": "PREDICTION, unverified: restated StdRng + oracle; compare with golden_gen.rs output",
           "seed_2321": run(2321, 8), "seed_0xbeef": run(0xBEEF, 4)}
    with open(os.path.join(HERE, "predicted_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote predicted_reference.json")


if __name__ == "__main__":
    main()
