"""GPU tests of the random-linear-combination fast accept (include/dsv.h:
dsv_verify_single_rlc_dev; schnorr_amd/csrc/k_rlc.hip).

The contract: the verdict vector of `PublicKey::verify` (/root/reference/src/keys/public.rs:121-130)
item by item — i.e. the ORACLE's — on every input; `accepted` says whether the aggregate decided.
An all-valid batch of prime-order points must be accepted (if the bucket sums, the per-bit subset
sums or the weights were wrong the aggregate would not be the identity), and a batch holding one
wrong signature, one point with a small-order component or one point off the curve must not be.
"""
import numpy as np
import pytest
import torch

import harness as H
import oracle_lib as O
import pymodel as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


COLS = {"single": ("u", "R", "PK", "m"), "double": ("u", "R", "Rp", "PK", "PKp", "m"),
        "vargen": ("u", "R", "PK", "Gen", "m")}


def _scheme_of(a):
    return "double" if "Rp" in a else ("vargen" if "Gen" in a else "single")


def _run(engine, a, window_bits=0):
    scheme = _scheme_of(a)
    n = len(a["u"])
    t = [torch.from_numpy(np.ascontiguousarray(a[k])).to(DEV) for k in COLS[scheme]]
    ok = torch.full((n,), 7, dtype=torch.uint8, device=DEV)
    ws = torch.empty(engine.rlc_workspace_bytes(n, window_bits), dtype=torch.uint8, device=DEV)
    accepted = getattr(engine, "verify_%s_rlc_dev" % scheme)(*t, ok, ws, window_bits=window_bits)
    torch.cuda.synchronize()
    return accepted, ok.cpu().numpy()


def _oracle(a):
    scheme = _scheme_of(a)
    return getattr(O, "verify_" + scheme)(*[a[k] for k in COLS[scheme]], nthreads=8)


def _signed(n, seed, scheme="single"):
    d = getattr(O, "keygen_sign_" + scheme)(n, seed, nthreads=8)
    return {k: d[k] for k in COLS[scheme]}


@pytest.mark.parametrize("scheme", ["double", "vargen"])
def test_double_and_var_generator_schemes(engine, scheme):
    """`PublicKeyDouble::verify` / `PublicKeyVarGen::verify` (src/keys/public.rs:222-244, :401-415) through
    the aggregate: all valid -> accepted; one wrong field anywhere -> the per-signature kernels' (= the
    oracle's) verdicts; the harness's tamper classes; a key / generator with a small-order component."""
    n = 1800
    d = _signed(n, 970 + len(scheme), scheme)
    for bits in (8, 12, 0):
        accepted, ok = _run(engine, d, bits)
        assert accepted == (bits != 0) and ok.all(), bits      # (automatic bits: too small for an aggregate)
    points = [k for k in COLS[scheme] if k not in ("u", "m")]
    for j, field in enumerate(COLS[scheme]):
        a = {k: v.copy() for k, v in d.items()}
        victim = 100 + 37 * j
        if field in ("u", "m"):
            a[field][victim, 2] ^= 0x04
        else:
            a[field][victim] = d[field][victim + 1]
        want = _oracle(a)
        assert want.sum() == n - 1 and not want[victim], field
        accepted, ok = _run(engine, a, 8)
        assert not accepted and np.array_equal(ok, want), field
    a = {k: v.copy() for k, v in d.items()}
    H.tamper(a, period=9)
    want = _oracle(a)
    assert 0 < want.sum() < n
    accepted, ok = _run(engine, a)
    assert not accepted and np.array_equal(ok, want)
    # small-order component in ONE point of one item, for every point column: never decided by the aggregate
    import test_halfgcd as TH
    t8 = TH.order8_point()
    for field in points:
        a = {k: v.copy() for k, v in d.items()}
        P = H.to_int_point(a[field][50])
        a[field][50] = np.frombuffer(M.point_bytes(M.padd(P, M.pmul(t8, 4))), np.uint8)
        want = _oracle(a)
        accepted, ok = _run(engine, a, 8)
        assert not accepted and np.array_equal(ok, want), field


@pytest.mark.parametrize("bits", [4, 6, 8, 12, 14, 16, 0])
def test_all_valid_batch_is_accepted_by_the_aggregate(engine, bits):
    d = _signed(1500, 900 + bits)
    if bits == 0:
        # automatic window bits: groups below 2^17 items skip the aggregate (it would be slower) ...
        accepted, ok = _run(engine, d)
        assert not accepted and ok.all()
        # ... from there on it decides
        reps = -(-((1 << 17) + 5) // 1500)
        d = {k: np.tile(v, (reps, 1))[:(1 << 17) + 5] for k, v in d.items()}
    accepted, ok = _run(engine, d, bits)
    assert accepted and ok.all()


@pytest.mark.parametrize("bits", [8, 12, 0])
def test_one_wrong_signature_sends_the_batch_to_the_per_signature_kernels(engine, bits):
    n = 2000
    d = _signed(n, 910 + bits)
    for victim, field in ((0, "u"), (n - 1, "m"), (n // 2, "PK"), (777, "R")):
        a = {k: v.copy() for k, v in d.items()}
        if field in ("u", "m"):
            a[field][victim, 3] ^= 0x10
        else:  # another point of the subgroup
            a[field][victim] = d[field][(victim + 1) % n]
        want = O.verify_single(a["u"], a["R"], a["PK"], a["m"], nthreads=8)
        assert want.sum() == n - 1 and not want[victim]
        accepted, ok = _run(engine, a, bits)
        assert not accepted
        assert np.array_equal(ok, want)


def test_tampered_batch_matches_the_oracle(engine):
    n = 6000
    d = _signed(n, 920)
    H.tamper(d, period=16)
    want = O.verify_single(d["u"], d["R"], d["PK"], d["m"], nthreads=8)
    assert 0 < want.sum() < n
    accepted, ok = _run(engine, d)
    assert not accepted and np.array_equal(ok, want)


def test_malformed_items_stay_out_of_the_sum(engine):
    """u >= r, a coordinate >= q, m >= q: verdict false by the encoding alone (the reference's types
    cannot hold them); the other items are all valid and the aggregate accepts them."""
    n = 1200
    d = _signed(n, 930)
    top = np.frombuffer(b"\xff" * 32, np.uint8)
    d["u"][5] = top
    d["R"][60, 32:] = top
    d["PK"][700, :32] = top
    d["m"][1100] = top
    want = O.verify_single(d["u"], d["R"], d["PK"], d["m"], nthreads=8)
    assert want.sum() == n - 4
    accepted, ok = _run(engine, d, 8)
    assert np.array_equal(ok, want)
    assert accepted


def _torsion_rows(t8, rnd, count, cancel):
    """signatures whose key and nonce point carry order-8 components; `cancel`: c*k1 == k2 (mod 8),
    i.e. VALID under the reference's cofactorless equation"""
    rows = {"u": [], "R": [], "PK": [], "m": []}
    while len(rows["u"]) < count:
        sk, m, rr = rnd.randrange(1, M.R_ORDER), rnd.randrange(M.Q), rnd.randrange(1, M.R_ORDER)
        k1, k2 = rnd.randrange(1, 8), rnd.randrange(8)
        pk = M.padd(M.pmul(M.GEN, sk), M.pmul(t8, k1))
        R = M.padd(M.pmul(M.GEN, rr), M.pmul(t8, k2))
        c = M.challenge(R, m)
        if ((c * k1 - k2) % 8 == 0) != cancel:
            continue
        rows["u"].append(np.frombuffer(M.le32((rr - c * sk) % M.R_ORDER), np.uint8))
        rows["R"].append(np.frombuffer(M.point_bytes(R), np.uint8))
        rows["PK"].append(np.frombuffer(M.point_bytes(pk), np.uint8))
        rows["m"].append(np.frombuffer(M.le32(m), np.uint8))
    return {k: np.stack(v) for k, v in rows.items()}


def test_small_order_components_are_never_decided_by_the_aggregate(engine):
    """The reference's equation is cofactorless: a key / nonce point with an order-8 component verifies
    exactly when the torsion parts cancel.  Whatever they do, a batch holding such a point must fall back
    — also when all its items are VALID (the aggregate's subgroup test is about the inputs, not the
    verdicts), and when two order-2 defects would cancel in a plain sum of the equations."""
    import test_halfgcd as TH
    t8, rnd = TH.order8_point(), TH.rnd
    n = 900
    base = _signed(n, 940)
    for cancel, count in ((True, 1), (False, 1), (True, 3), (False, 2)):
        rows = _torsion_rows(t8, rnd, count, cancel)
        a = {k: v.copy() for k, v in base.items()}
        for j in range(count):
            for k in rows:
                a[k][100 + 250 * j] = rows[k][j]
        want = O.verify_single(a["u"], a["R"], a["PK"], a["m"], nthreads=8)
        assert want.sum() == (n if cancel else n - count)
        accepted, ok = _run(engine, a, 8)
        assert not accepted, (cancel, count)
        assert np.array_equal(ok, want)
    # two items with the SAME order-2 defect: R + T2 signed as if it were R (T2 = (0, -1), so R + T2 =
    # (-u, -v)); z_a D + z_b D vanishes in a plain weighted sum whenever z_a + z_b is even
    t2 = M.pmul(t8, 4)
    assert t2 == (0, M.Q - 1)
    a = {k: v.copy() for k, v in base.items()}
    for i in (10, 20):
        sk, m, rr = rnd.randrange(1, M.R_ORDER), rnd.randrange(M.Q), rnd.randrange(1, M.R_ORDER)
        R = M.padd(M.pmul(M.GEN, rr), t2)
        c = M.challenge(R, m)
        a["u"][i] = np.frombuffer(M.le32((rr - c * sk) % M.R_ORDER), np.uint8)
        a["R"][i] = np.frombuffer(M.point_bytes(R), np.uint8)
        a["PK"][i] = np.frombuffer(M.point_bytes(M.pmul(M.GEN, sk)), np.uint8)
        a["m"][i] = np.frombuffer(M.le32(m), np.uint8)
    want = O.verify_single(a["u"], a["R"], a["PK"], a["m"], nthreads=8)
    assert want.sum() == n - 2
    for _ in range(6):  # fresh weights every call
        accepted, ok = _run(engine, a, 8)
        assert not accepted and np.array_equal(ok, want)


def test_point_off_the_curve_takes_the_per_signature_path(engine):
    n = 800
    d = _signed(n, 950)
    d["PK"][33, 0] ^= 1  # canonical coordinates, not on the curve (the reference's types cannot hold it)
    t = {k: torch.from_numpy(d[k]).to(DEV) for k in ("u", "R", "PK", "m")}
    ok0 = torch.zeros(n, dtype=torch.uint8, device=DEV)
    ws0 = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device=DEV)
    engine.verify_single_dev(t["u"], t["R"], t["PK"], t["m"], ok0, ws0)
    torch.cuda.synchronize()
    accepted, ok = _run(engine, d, 8)
    assert not accepted
    assert np.array_equal(ok, ok0.cpu().numpy())


def test_argument_checks(engine):
    from schnorr_amd import _lib
    d = _signed(16, 960)
    with pytest.raises(ValueError):
        engine.rlc_workspace_bytes(16, 7)
    t = {k: torch.from_numpy(d[k]).to(DEV) for k in ("u", "R", "PK", "m")}
    ok = torch.zeros(16, dtype=torch.uint8, device=DEV)
    ws = torch.empty(engine.rlc_workspace_bytes(16), dtype=torch.uint8, device=DEV)
    import ctypes
    L = _lib.load()
    rc = L.dsv_verify_single_rlc_dev(ctypes.c_void_p(t["u"].data_ptr()), ctypes.c_void_p(t["R"].data_ptr()),
                                     ctypes.c_void_p(t["PK"].data_ptr()), ctypes.c_void_p(t["m"].data_ptr()),
                                     ctypes.c_size_t(16), ctypes.c_void_p(ok.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                     None, ctypes.c_int(18), None)
    assert rc == -2
    accepted, got = _run(engine, d)  # tiny batch, default bits: the per-signature kernels
    assert not accepted and got.all()
    accepted, got = _run(engine, d, 4)
    assert accepted and got.all()
    acc0 = ctypes.c_int(5)
    rc = L.dsv_verify_single_rlc_dev(None, None, None, None, ctypes.c_size_t(0), None, None, None, ctypes.c_int(0),
                                     ctypes.byref(acc0))
    assert rc == 0


def test_full_size_batches(engine):
    """2^20 signatures (BASELINE configs[1]'s size): all valid -> accepted; the graded workload (1/16
    tampered) -> the per-signature kernels' verdicts, equal to the construction-time pattern."""
    from schnorr_amd import workload as W
    n = 1 << 20
    ws = torch.empty(engine.rlc_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    ok = torch.zeros(n, dtype=torch.uint8, device=DEV)
    b = W.gen_single(n, seed=2321, tamper=False)
    assert engine.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    assert bool(ok.all())
    b = W.gen_single(n, seed=2321)
    ok.zero_()
    assert not engine.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    assert torch.equal(ok, b["expected"])


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_host_fast_accept_over_typed_objects(engine, scheme):
    """dsv_verify_*_mont_cols_rlc: records laid out like the Rust structs (Montgomery limbs, projective
    points with a random z; tests/mont_cases.py) in host memory.  The oracle's verdicts on a tampered
    batch (per-signature kernels on the resident arena), on its valid items alone (accepted) and on valid
    + malformed items — z = 0, limbs >= the modulus: verdict 0 by the encoding, out of the sum — (accepted);
    a one-chunk call (too small for an aggregate: the ordinary path) and one of several chunks with a ragged tail."""
    import mont_cases as C
    cols, want = C.mont_case(scheme, 400, 980 + len(scheme), period=5)
    assert 0 < want.sum() < len(want)
    for n in (313, (1 << 17) + (1 << 15) + 77):      # (below 2^17 items: the ordinary column path)
        reps = -(-n // 400)
        tcols = [np.ascontiguousarray(np.tile(c, (reps, 1))[:n]) for c in cols]
        got, accepted = engine.verify_mont_cols_rlc(scheme, C.as_records(scheme, tcols)[3])
        assert not accepted and np.array_equal(got, np.tile(want, reps)[:n]), n
        keep = np.flatnonzero(want)
        reps = -(-n // len(keep))
        vcols = [np.ascontiguousarray(np.tile(c[keep], (reps, 1))[:n]) for c in cols]
        got, accepted = engine.verify_mont_cols_rlc(scheme, C.as_records(scheme, vcols)[3])
        assert accepted == (n >= 1 << 17) and got.all(), n
    # only item 0 tampered (dropped); the planted encodings the Rust types cannot hold stay in
    cols, want = C.mont_case(scheme, 300, 990 + len(scheme), period=10 ** 9)
    assert not want[0] and want[1:].sum() == 299 - 2 * (len(cols) - 2) - 2
    n = (1 << 17) + 5
    reps = -(-n // 299)
    mcols = [np.ascontiguousarray(np.tile(c[1:], (reps, 1))[:n]) for c in cols]
    got, accepted = engine.verify_mont_cols_rlc(scheme, C.as_records(scheme, mcols)[3])
    assert accepted and np.array_equal(got, np.tile(want[1:], reps)[:n])
    # the same objects through the ordinary column path: identical verdicts
    assert np.array_equal(engine.verify_mont_cols(scheme, C.as_records(scheme, mcols)[3]), got)


def test_host_fast_accept_shards_like_the_multi_forms(engine, monkeypatch):
    """DSV_MULTI_SHARDS=3 on one GPU: three shards, each ONE group with its own aggregate (they take the
    device's arena in turn); accepted only if every shard's aggregate accepted."""
    import mont_cases as C
    cols, want = C.mont_case("single", 300, 1001, period=10 ** 9)   # item 0 tampered, planted encodings
    n = 3 * (1 << 17) + 11
    reps = -(-n // 299)
    good = [np.ascontiguousarray(np.tile(c[1:], (reps, 1))[:n]) for c in cols]
    gwant = np.tile(want[1:], reps)[:n]
    monkeypatch.setenv("DSV_MULTI_SHARDS", "3")
    got, accepted = engine.verify_mont_cols_rlc("single", C.as_records("single", good)[3])
    assert accepted and np.array_equal(got, gwant)
    bad = [c.copy() for c in good]
    victim = n - 5                                   # in the last shard only
    assert gwant[victim]
    bad[0][victim] = good[0][victim - 1]             # another item's u
    bwant = gwant.copy()
    bwant[victim] = 0
    got, accepted = engine.verify_mont_cols_rlc("single", C.as_records("single", bad)[3])
    assert not accepted and np.array_equal(got, bwant)


def test_more_than_one_group(engine):
    """2^22 + 777 signatures: two groups of half the batch each, each with its own aggregate and weights.
    All valid -> accepted; one wrong signature in the second group -> only that group falls back, the
    verdicts are the construction-time pattern either way."""
    from schnorr_amd import workload as W
    n = (1 << 22) + 777
    b = W.gen_single(n, seed=5, tamper=False)
    ws = torch.empty(engine.rlc_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    ok = torch.zeros(n, dtype=torch.uint8, device=DEV)
    assert engine.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    assert bool(ok.all())
    b["u"][n - 3] = b["u"][n - 4]
    ok.zero_()
    assert not engine.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    want = torch.ones(n, dtype=torch.uint8, device=DEV)
    want[n - 3] = 0
    assert torch.equal(ok, want)


def test_host_fast_accept_from_several_threads(engine):
    """Three threads on the blocking typed-object entry points at once, a different scheme each (two
    arenas per device: two calls overlap, the third waits), valid and tampered batches alternating; beside
    them a thread on the ordinary column path.  Every call's verdicts are the oracle's."""
    import threading
    import mont_cases as C
    n = (1 << 17) + (1 << 14) + 9
    work = []
    for scheme in ("single", "double", "vargen"):
        cols, want = C.mont_case(scheme, 300, 1010 + len(scheme), period=6)
        reps = -(-n // 300)
        bad = C.as_records(scheme, [np.ascontiguousarray(np.tile(c, (reps, 1))[:n]) for c in cols])[3]
        keep = np.flatnonzero(want)
        reps = -(-n // len(keep))
        good = C.as_records(scheme, [np.ascontiguousarray(np.tile(c[keep], (reps, 1))[:n]) for c in cols])[3]
        work.append((scheme, bad, np.tile(want, -(-n // 300))[:n], good))
    errors = []

    def fast(scheme, bad, bwant, good):
        try:
            for rep in range(3):
                got, acc = engine.verify_mont_cols_rlc(scheme, good)
                if not acc or not got.all():
                    errors.append("%s valid batch, call %d: accepted=%d" % (scheme, rep, acc))
                got, acc = engine.verify_mont_cols_rlc(scheme, bad)
                if acc or not np.array_equal(got, bwant):
                    errors.append("%s tampered batch, call %d" % (scheme, rep))
        except Exception as e:  # noqa: BLE001
            errors.append("%s: %r" % (scheme, e))

    def plain():
        try:
            for _ in range(4):
                if not np.array_equal(engine.verify_mont_cols(work[0][0], work[0][1]), work[0][2]):
                    errors.append("ordinary column path beside the fast accept: verdicts differ")
        except Exception as e:  # noqa: BLE001
            errors.append("plain: %r" % (e,))

    th = [threading.Thread(target=fast, args=w) for w in work] + [threading.Thread(target=plain)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_wire_records_through_the_fast_accept(engine, scheme):
    """dsv_verify_*_wire_rlc_dev against the oracle's from_bytes + verify: a tampered batch with undecodable
    records (per-signature kernels decide), and valid + undecodable records alone (the aggregate decides:
    what does not decode has verdict 0 and stays out of the sum)."""
    n = 700
    d = _signed(n, 1020 + len(scheme), scheme)
    cp = engine.compress_points
    if scheme == "single":
        sig, pk = np.concatenate([d["u"], cp(d["R"])], axis=1), cp(d["PK"])
    elif scheme == "double":
        sig = np.concatenate([d["u"], cp(d["R"]), cp(d["Rp"])], axis=1)
        pk = np.concatenate([cp(d["PK"]), cp(d["PKp"])], axis=1)
    else:
        sig, pk = np.concatenate([d["u"], cp(d["R"])], axis=1), np.concatenate([cp(d["PK"]), cp(d["Gen"])], axis=1)
    sig, pk, m = np.ascontiguousarray(sig), np.ascontiguousarray(pk), d["m"].copy()
    for bit in range(64):    # a compressed R that decodes to NO curve point (about half of all v do not)
        cand = sig[5:6, 32:64].copy()
        cand[0, bit >> 3] ^= 1 << (bit & 7)
        if not O.decompress(cand)[1][0]:
            sig[5, 32:64] = cand[0]
            break
    else:
        raise AssertionError("no undecodable neighbour found")
    pk[7, -1] |= 0x7f        # v >= q
    wire = getattr(O, "verify_%s_wire" % scheme)

    def run(sig, pk, m, bits):
        t = [torch.from_numpy(np.ascontiguousarray(x)).to(DEV) for x in (sig, pk, m)]
        ok = torch.full((len(m),), 5, dtype=torch.uint8, device=DEV)
        ws = torch.empty(engine.wire_rlc_workspace_bytes(len(m), bits), dtype=torch.uint8, device=DEV)
        acc = engine.verify_wire_rlc_dev(scheme, *t, ok, ws, window_bits=bits)
        torch.cuda.synchronize()
        return acc, ok.cpu().numpy()

    want = wire(sig, pk, m)
    assert want.sum() == n - 2 and not want[5] and not want[7]
    for bits in (8, 12):
        acc, ok = run(sig, pk, m, bits)
        assert acc and np.array_equal(ok, want), bits
    m2 = m.copy()
    m2[300, 1] ^= 2
    want2 = wire(sig, pk, m2)
    assert want2.sum() == n - 3
    acc, ok = run(sig, pk, m2, 8)
    assert not acc and np.array_equal(ok, want2)
    # automatic bits past 2^17 records
    big = (1 << 17) + 3
    reps = -(-big // n)
    tile = lambda a: np.tile(a, (reps, 1))[:big]
    acc, ok = run(tile(sig), tile(pk), tile(m), 0)
    assert acc and np.array_equal(ok, np.tile(want, reps)[:big])
    acc, ok = run(tile(sig), tile(pk), tile(m2), 0)
    assert not acc and np.array_equal(ok, np.tile(want2, reps)[:big])
    # the same records from HOST memory (dsv_verify_*_wire_rlc): decoded chunk by chunk into the arena
    ok, acc = engine.verify_wire_rlc(scheme, tile(sig), tile(pk), tile(m))
    assert acc and np.array_equal(ok, np.tile(want, reps)[:big])
    ok, acc = engine.verify_wire_rlc(scheme, tile(sig), tile(pk), tile(m2))
    assert not acc and np.array_equal(ok, np.tile(want2, reps)[:big])
    ok, acc = engine.verify_wire_rlc(scheme, sig, pk, m)          # too small for an aggregate: the ordinary path
    assert not acc and np.array_equal(ok, want)


def test_sample_check_follows_the_recent_groups():
    """The adaptive sample check (dsv_host.h: Context::rlc_suspicion), seen through DSV_RLC_TRACE in a process of its
    own: valid batches switch it off; the next tampered batch then pays its aggregate ("sum"), which switches
    it on; the one after that is caught by the sample ("no aggregate").  Verdicts are the pattern every time."""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np, torch
from schnorr_amd import engine as E, workload as W
E.init(0)
n = (1 << 17) + 64
ws = torch.empty(E.rlc_workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
good = W.gen_single(n, seed=3, tamper=False)
bad = W.gen_single(n, seed=3, tamper=True)          # every 16th item wrong: some among the first 1024
run = lambda b: E.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
for label, b, want_acc in (("valid-1", good, True), ("valid-2", good, True), ("tampered-1", bad, False),
                           ("tampered-2", bad, False), ("valid-3", good, True)):
    print("CALL", label, flush=True)
    import sys; sys.stderr.write("CALL %s\n" % label); sys.stderr.flush()
    acc = run(b)
    assert acc == want_acc and torch.equal(ok, b["expected"]), label
print("done")
"""
    env = dict(os.environ, DSV_RLC_TRACE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    calls = {}
    cur = None
    for line in r.stderr.splitlines():
        if line.startswith("CALL "):
            cur = line[5:]
            calls[cur] = []
        elif "[dsv rlc]" in line and cur:
            calls[cur].append(line)
    assert any("accepted" in x for x in calls["valid-1"]) and any("accepted" in x for x in calls["valid-2"])
    assert any(" sum" in x for x in calls["tampered-1"]) and not any("no aggregate" in x for x in calls["tampered-1"])
    assert any("no aggregate" in x for x in calls["tampered-2"])
    assert any("accepted" in x for x in calls["valid-3"])          # the sample passes, the aggregate decides


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_host_fast_accept_with_the_bucket_pass_in_two_ranges(engine, scheme):
    """From 2^18 items on the host forms run their bucket pass in two ranges (dsv_rlc.hip: RlcHook): the first
    while the second is still on the bus, then the second, a merge of the two bucket arrays and the tail.
    Valid + malformed items -> accepted; ONE wrong signature in the first range, in the second, in the item at
    the range boundary's neighbourhood -> the per-signature kernels' (the oracle's) verdicts, not accepted.
    Typed objects and serialized records."""
    import mont_cases as C
    n = (1 << 18) + (1 << 16) + 5
    cols, want = C.mont_case(scheme, 300, 1030 + len(scheme), period=10 ** 9)   # item 0 tampered (dropped), planted encodings kept
    reps = -(-n // 299)
    good = [np.ascontiguousarray(np.tile(c[1:], (reps, 1))[:n]) for c in cols]
    gwant = np.tile(want[1:], reps)[:n]
    got, accepted = engine.verify_mont_cols_rlc(scheme, C.as_records(scheme, good)[3])
    assert accepted and np.array_equal(got, gwant)
    for victim in (7, n // 2 - 3, n // 2 + (1 << 15), n - 2):
        while not gwant[victim]:
            victim += 1
        bad = [c.copy() for c in good]
        src = victim - 1 if gwant[victim - 1] else victim + 1
        bad[0][victim] = good[0][src]                        # another item's u
        bwant = gwant.copy()
        bwant[victim] = 0
        got, accepted = engine.verify_mont_cols_rlc(scheme, C.as_records(scheme, bad)[3])
        assert not accepted and np.array_equal(got, bwant), victim
    # serialized records
    d = _signed(500, 1040 + len(scheme), scheme)
    cp = engine.compress_points
    if scheme == "single":
        sig, pk = np.concatenate([d["u"], cp(d["R"])], axis=1), cp(d["PK"])
    elif scheme == "double":
        sig = np.concatenate([d["u"], cp(d["R"]), cp(d["Rp"])], axis=1)
        pk = np.concatenate([cp(d["PK"]), cp(d["PKp"])], axis=1)
    else:
        sig, pk = np.concatenate([d["u"], cp(d["R"])], axis=1), np.concatenate([cp(d["PK"]), cp(d["Gen"])], axis=1)
    reps = -(-n // 500)
    tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    tsig, tpk, tm = tile(sig), tile(pk), tile(d["m"])
    got, accepted = engine.verify_wire_rlc(scheme, tsig, tpk, tm)
    assert accepted and got.all()
    for victim in (3, n - 9):
        m2 = tm.copy()
        m2[victim, 0] ^= 1
        got, accepted = engine.verify_wire_rlc(scheme, tsig, tpk, m2)
        assert not accepted and got.sum() == n - 1 and not got[victim]


def test_mixed_batch_through_the_fast_accept(engine):
    """dsv_verify_mixed_rlc_dev (BASELINE configs[4]'s shape: singles and doubles interleaved): each kind's items
    are one group.  All valid -> accepted; one wrong double signature -> only that kind's group falls back,
    not accepted; a wrong n_double -> every verdict 0, as dsv_verify_mixed_dev."""
    from schnorr_amd import workload as W
    n = 1 << 19                                        # 2^18 singles + 2^18 doubles
    b = W.gen_mixed(n, seed=77, tamper=False)
    ws = torch.empty(engine.mixed_rlc_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    ok = torch.zeros(n, dtype=torch.uint8, device=DEV)
    args = lambda bb: (bb["kinds"], bb["u"], bb["R"], bb["Rp"], bb["PK"], bb["PKp"], bb["m"])
    assert engine.verify_mixed_rlc_dev(*args(b), b["n_double"], ok, ws)
    torch.cuda.synchronize()
    assert bool(ok.all())
    b["u"][n - 1] = b["u"][n - 3]                      # the last double item
    ok.zero_()
    assert not engine.verify_mixed_rlc_dev(*args(b), b["n_double"], ok, ws)
    torch.cuda.synchronize()
    want = torch.ones(n, dtype=torch.uint8, device=DEV)
    want[n - 1] = 0
    assert torch.equal(ok, want)
    t = W.gen_mixed(n, seed=78)                        # the graded pattern in both kinds: the plain entry point's verdicts
    ok.zero_()
    assert not engine.verify_mixed_rlc_dev(*args(t), t["n_double"], ok, ws)
    torch.cuda.synchronize()
    assert torch.equal(ok, t["expected"])
    ok.fill_(3)
    engine.verify_mixed_rlc_dev(*args(t), t["n_double"] - 1, ok, ws)   # a wrong count
    torch.cuda.synchronize()
    assert not bool(ok.any())
