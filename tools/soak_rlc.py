#!/usr/bin/env python3
"""Soak of the batch fast accept (dsv_verify_*_rlc_dev) against the oracle, bit for bit:

    python tools/soak_rlc.py [rounds=60] [seed=1]

Every round: one scheme, a size between 40 and 2^17 + change (so that every default window width and
the group logic see ragged sizes), explicit or automatic window bits, and one of
  clean      all valid                               -> must be ACCEPTED, all verdicts 1
  one        one wrong field in one random item      -> must NOT be accepted
  few        the harness's tamper classes, sparse    -> must NOT be accepted
  torsion    an order-8 / order-4 / order-2 component added to one random point (the signature may
             stay valid under the reference's cofactorless equation or not) -> must NOT be accepted
  malformed  non-canonical encodings only            -> accepted, those items 0
Every fourth round runs the TYPED-OBJECT form instead (dsv_verify_*_mont_cols_rlc over records laid out like the
Rust structs, Montgomery limbs, random z, planted encodings the types cannot hold: tests/mont_cases.py):
tampered -> not accepted; its valid + malformed items alone -> accepted.  Every eighth round: the batch as
serialized records in host memory (dsv_verify_*_wire_rlc) against the oracle's from_bytes + verify.
r06: every round also draws the sub-group count (automatic, or 1 .. 16 forced: dsv_debug_rlc_subgroups) and the
device's history counters (short 0: one sub-group, no sample; 8: sub-groups + the sample check; long 128 with short 0:
guarded groups, a gated second stage of sub-group aggregates), and every third
device-pointer round takes `accepted` through a pinned word (the enqueue-only form) instead of a host int.
Verdicts always equal the oracle's (the oracle is test infrastructure; nothing here is timed)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import harness as H  # noqa: E402
import mont_cases as C  # noqa: E402
import oracle_lib as O  # noqa: E402
import pymodel as M  # noqa: E402
import test_halfgcd as TH  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402

COLS = {"single": ("u", "R", "PK", "m"), "double": ("u", "R", "Rp", "PK", "PKp", "m"),
        "vargen": ("u", "R", "PK", "Gen", "m")}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
E.init(0)
t8 = TH.order8_point()
SIZES = [40, 700, 4097, (1 << 14) + 3, (1 << 15) - 1, 40000, (1 << 16) + 77, (1 << 17) + 1, (1 << 17) + (1 << 16) + 13, (1 << 18) + 5, (1 << 18) + (1 << 17) + 7]
total = bad = 0
t0 = time.time()
for rd in range(rounds):
    scheme = ("single", "double", "vargen")[int(rng.integers(0, 3))]
    n = int(SIZES[int(rng.integers(0, len(SIZES)))])
    forced = int((0, 0, 0, 1, 2, 3, 4, 8, 16)[int(rng.integers(0, 9))])
    hist = int((0, 8)[int(rng.integers(0, 2))])
    hist_long = int((0, 128)[int(rng.integers(0, 2))])   # (with hist 0: "guarded" groups, a gated second stage)
    E.rlc_subgroups(forced)
    E.rlc_history(0, hist)
    E.rlc_history_long(0, hist_long)
    if rd % 4 == 3:
        cols, want = C.mont_case(scheme, 300, int(rng.integers(1, 1 << 30)), period=int(rng.integers(4, 40)))
        tamper_free = bool(rng.integers(0, 2))
        if tamper_free:   # keep the valid items and the planted malformed ones (verdict 0 by the encoding)
            bad_enc = np.zeros(300, bool)   # items whose verdict is 0 by their encoding alone
            for k, c in enumerate(cols):
                width = c.shape[1]
                for j in range(0, width, 32):
                    mod = C.R_ORDER if k == 0 else C.Q
                    vals = [int.from_bytes(bytes(row[j:j + 32]), "little") for row in c]
                    bad_enc |= np.array([v >= mod for v in vals])
                if 0 < k < len(cols) - 1:
                    bad_enc |= np.array([not any(row[64:96]) for row in c])
            keep = (want == 1) | bad_enc
            cols = [c[keep] for c in cols]
            want = want[keep]
        base = len(want)
        reps = -(-n // base)
        tcols = [np.ascontiguousarray(np.tile(c, (reps, 1))[:n]) for c in cols]
        twant = np.tile(want, reps)[:n]
        got, accepted = E.verify_mont_cols_rlc(scheme, C.as_records(scheme, tcols)[3])
        diff = int((got != twant).sum())
        wrong_accept = accepted != (tamper_free and n >= 1 << 17)   # (smaller batches take the ordinary path)
        total += n
        bad += diff + (1 if wrong_accept else 0)
        print("round %d: %s n=%d typed objects %-9s accepted=%d valid=%d/%d%s  (%.0f s)" % (
            rd, scheme, n, "clean+enc" if tamper_free else "tampered", accepted, int(twant.sum()), n,
            "  DIFFERENT: %d verdicts%s" % (diff, ", acceptance" if wrong_accept else "") if diff or wrong_accept else "",
            time.time() - t0), flush=True)
        continue
    bits = int((0, 0, 0, 4, 6, 8, 12, 14, 16)[int(rng.integers(0, 9))])
    if bits and bits < 8 and n > 5000:
        bits = 8
    kind = ("clean", "one", "few", "torsion", "malformed")[int(rng.integers(0, 5))]
    base = min(n, 1500)
    d = getattr(O, "keygen_sign_" + scheme)(base, int(rng.integers(1, 1 << 30)), nthreads=8)
    d = {k: d[k] for k in COLS[scheme]}
    points = [k for k in COLS[scheme] if k not in ("u", "m")]
    if kind == "one":
        f = COLS[scheme][int(rng.integers(0, len(COLS[scheme])))]
        i = int(rng.integers(0, base))
        if f in ("u", "m"):
            d[f][i, int(rng.integers(0, 31))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        else:
            d[f][i] = d[f][(i + 1) % base].copy()
    elif kind == "few":
        H.tamper(d, period=int(rng.integers(50, 400)))
    elif kind == "torsion":
        f = points[int(rng.integers(0, len(points)))]
        i = int(rng.integers(0, base))
        P = H.to_int_point(d[f][i])
        d[f][i] = np.frombuffer(M.point_bytes(M.padd(P, M.pmul(t8, int(rng.integers(1, 8))))), np.uint8)
    elif kind == "malformed":
        top = np.frombuffer(b"\xff" * 32, np.uint8)
        for _ in range(3):
            f = COLS[scheme][int(rng.integers(0, len(COLS[scheme])))]
            i = int(rng.integers(0, base))
            if f in ("u", "m"):
                d[f][i] = top
            else:
                d[f][i, 32 * int(rng.integers(0, 2)):][:32] = top
    want = getattr(O, "verify_" + scheme)(*[d[k] for k in COLS[scheme]], nthreads=8)
    if rd % 8 == 5 and kind != "malformed":
        # the same batch as serialized records in host memory (dsv_verify_*_wire_rlc); a point with a small-order
        # component survives compression, so every defect kind carries over
        cp = E.compress_points
        if scheme == "single":
            sig, pk = np.concatenate([d["u"], cp(d["R"])], axis=1), cp(d["PK"])
        elif scheme == "double":
            sig = np.concatenate([d["u"], cp(d["R"]), cp(d["Rp"])], axis=1)
            pk = np.concatenate([cp(d["PK"]), cp(d["PKp"])], axis=1)
        else:
            sig, pk = np.concatenate([d["u"], cp(d["R"])], axis=1), np.concatenate([cp(d["PK"]), cp(d["Gen"])], axis=1)
        wwant = getattr(O, "verify_%s_wire" % scheme)(np.ascontiguousarray(sig), np.ascontiguousarray(pk), d["m"])
        reps = -(-n // base)
        tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
        got, accepted = E.verify_wire_rlc(scheme, tile(sig), tile(pk), tile(d["m"]))
        twant = np.tile(wwant, reps)[:n]
        diff = int((got != twant).sum())
        # (a point with a small-order component may leave its item valid: never accepted all the same)
        wrong_accept = accepted != (kind == "clean" and n >= 1 << 17)
        total += n
        bad += diff + (1 if wrong_accept else 0)
        print("round %d: %s n=%d wire records %-9s accepted=%d valid=%d/%d%s  (%.0f s)" % (
            rd, scheme, n, kind, accepted, int(twant.sum()), n,
            "  DIFFERENT: %d verdicts%s" % (diff, ", acceptance" if wrong_accept else "") if diff or wrong_accept else "",
            time.time() - t0), flush=True)
        continue
    # tile to n: items repeat, their weights do not (one z per item and call)
    reps = -(-n // base)
    a = [np.ascontiguousarray(np.tile(d[k], (reps, 1))[:n]) for k in COLS[scheme]]
    twant = np.tile(want, reps)[:n]
    t = [torch.from_numpy(x).to("cuda:0") for x in a]
    ok = torch.full((n,), 9, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(E.rlc_workspace_bytes(n, bits), dtype=torch.uint8, device="cuda:0")
    if rd % 3 == 1:
        word = torch.full((1,), 7, dtype=torch.int32).pin_memory()
        assert getattr(E, "verify_%s_rlc_dev" % scheme)(*t, ok, ws, window_bits=bits, accepted_out=word) is None
        torch.cuda.synchronize()
        accepted = bool(int(word[0]))
    else:
        accepted = getattr(E, "verify_%s_rlc_dev" % scheme)(*t, ok, ws, window_bits=bits)
    got = ok.cpu().numpy()
    # automatic bits: small groups skip the aggregate (rlc.h: rlc_min_auto — 2^17 single, 2^14 double / var-generator)
    expect_accept = kind in ("clean", "malformed") and (bits != 0 or n >= (1 << 17 if scheme == "single" else 1 << 14))
    diff = int((got != twant).sum())
    wrong_accept = accepted != expect_accept
    total += n
    bad += diff + (1 if wrong_accept else 0)
    print("round %d: %s n=%d bits=%d sub-groups=%s history=%d/%d %-9s accepted=%d valid=%d/%d%s  (%.0f s)" % (
        rd, scheme, n, bits, forced or "auto", hist, hist_long, kind, accepted, int(twant.sum()), n,
        "  DIFFERENT: %d verdicts%s" % (diff, ", acceptance" if wrong_accept else "") if diff or wrong_accept else "",
        time.time() - t0), flush=True)
    del ws, ok, t
E.rlc_subgroups(0)
E.rlc_history_long(0, 0)
print("soak_rlc: %d verdicts compared with the oracle, %d different / wrongly accepted" % (total, bad))
sys.exit(1 if bad else 0)
