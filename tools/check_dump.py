#!/usr/bin/env python3
"""Stand-in for rust/dusk-schnorr-gpu/src/bin/bench_ref.rs where there is no Rust toolchain: reads the
directory `bench.py --dump-inputs DIR` wrote (the timed batches as the reference's wire records + the GPU
verdicts), runs the CPU ORACLE's wire-format verify over them (tests/oracle_lib.py — test infrastructure,
a restatement of the reference, NOT the crate) and prints a JSON object of bench_ref's schema with
`"kind": "port"`.  Proves that the dump is what `Signature::from_bytes` / `PublicKey::from_bytes` /
`BlsScalar::from_bytes` expect (64 / 96 / 64-byte signatures, 32 / 64 / 64-byte keys: /root/reference
src/signatures.rs:106-123, 245-270, 387-404, src/keys/public.rs:87-101, 282-299, 347-372) and that
`bench.py --cpu-baseline-file` parses the schema (`--as-crate` labels the output "crate" for that test
only — never commit such a file as a measurement).

    python tools/check_dump.py DIR [--items N] [--as-crate]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402


def main():
    d = sys.argv[1]
    items = int(sys.argv[sys.argv.index("--items") + 1]) if "--items" in sys.argv else 1 << 13
    shapes = {"single": (64, 32, O.verify_single_wire), "double": (96, 64, O.verify_double_wire),
              "vargen": (64, 64, O.verify_vargen_wire)}
    out = {"kind": "crate" if "--as-crate" in sys.argv else "port",
           "crate": "oracle/schnorr_oracle.c (restatement; NOT dusk-schnorr)", "cores": 1, "cpu": ""}
    bad = 0
    for name, (sw, pw, fn) in shapes.items():
        f = lambda s: os.path.join(d, "%s_%s.bin" % (name, s))
        if not os.path.exists(f("sig")):
            continue
        exp = np.fromfile(f("expected"), dtype=np.uint8)
        n = exp.shape[0]
        sig = np.fromfile(f("sig"), dtype=np.uint8).reshape(n, sw)
        pk = np.fromfile(f("pk"), dtype=np.uint8).reshape(n, pw)
        m = np.fromfile(f("m"), dtype=np.uint8).reshape(n, 32)
        k = min(n, items)
        t0 = time.perf_counter()
        got = fn(sig[:k], pk[:k], m[:k])
        dt = time.perf_counter() - t0
        mism = int((got != exp[:k]).sum())
        bad += mism
        one = {"items": k, "seconds": dt, "value": k / dt}
        out[name] = {"items": n, "undecodable": None, "decode_seconds": None, "mismatches_vs_gpu": mism,
                     "threads_1": one, "threads_all": dict(one, threads=1)}
    print(json.dumps(out))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
