"""CPU tests of the oracle (oracle/schnorr_oracle.c): constants re-derived with Python integers,
agreement with the independent big-int model, the reference's own relational tests
(tests/schnorr.rs, schnorr_double.rs, schnorr_var_generator.rs, keys.rs), golden fixtures."""
import json
import os

import numpy as np
import pytest

import harness as H
import oracle_lib as O
import pymodel as M

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
unhex = lambda s: np.frombuffer(bytes.fromhex(s), dtype=np.uint8)


def _is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def test_field_and_curve_constants():
    """SURVEY.md Appendix A.1-A.3, marked [V]: primes, d, generators on-curve with order r."""
    assert _is_prime(M.Q) and _is_prime(M.R_ORDER)
    assert M.Q.bit_length() == 255 and M.R_ORDER.bit_length() == 252
    assert (M.Q - 1) % (1 << 32) == 0 and (M.Q - 1) % (1 << 33) != 0  # 2-adicity 32
    assert M.D == (-10240 * pow(10241, -1, M.Q)) % M.Q
    assert pow(M.D, (M.Q - 1) // 2, M.Q) == M.Q - 1  # d is a non-square: complete formulas
    for g in (M.GEN, M.GEN_NUMS):
        assert M.on_curve(g)
        assert M.pmul(g, M.R_ORDER) == M.IDENTITY
        assert M.pmul(g, 8) != M.IDENTITY
    assert M.compress(M.GEN).hex() == "12" + "00" * 31
    assert M.compress(M.GEN_NUMS).hex() == \
        "f83e2e1607b705677a50a5820fba4999fd343bebbe2d167b1bebf3b2b30ed8c3"


def test_oracle_matches_python_model_on_sign_and_hash():
    d = O.keygen_sign_single(5, 2321)
    c = O.challenge_single(d["R"], d["m"])
    for i in range(5):
        sk, m = M.from_le(d["sk"][i]), M.from_le(d["m"][i])
        R, PK = H.to_int_point(d["R"][i]), H.to_int_point(d["PK"][i])
        assert M.pmul(M.GEN, sk) == PK
        assert M.challenge(R, m) == M.from_le(c[i])
        # u = r - c*sk  =>  r*G == R; recover r from u
        r = (M.from_le(d["u"][i]) + M.from_le(c[i]) * sk) % M.R_ORDER
        assert M.pmul(M.GEN, r) == R
        assert M.sign_single(sk, m, r) == (M.from_le(d["u"][i]), R)


def test_sign_verify_wrong_key_single():  # tests/schnorr.rs:14-40
    d = O.keygen_sign_single(16, 2321, nthreads=4)
    assert O.verify_single(d["u"], d["R"], d["PK"], d["m"]).all()
    wrong = np.roll(d["PK"], 1, axis=0)
    assert not O.verify_single(d["u"], d["R"], wrong, d["m"]).any()


def test_sign_verify_wrong_key_double():  # tests/schnorr_double.rs:14-41
    d = O.keygen_sign_double(8, 2321, nthreads=4)
    assert O.verify_double(d["u"], d["R"], d["Rp"], d["PK"], d["PKp"], d["m"]).all()
    assert not O.verify_double(d["u"], d["R"], d["Rp"], np.roll(d["PK"], 1, 0),
                               np.roll(d["PKp"], 1, 0), d["m"]).any()
    # only the primed key wrong
    assert not O.verify_double(d["u"], d["R"], d["Rp"], d["PK"], np.roll(d["PKp"], 1, 0),
                               d["m"]).any()


def test_sign_verify_wrong_key_vargen():  # tests/schnorr_var_generator.rs:14-40
    d = O.keygen_sign_vargen(8, 2321, nthreads=4)
    assert O.verify_vargen(d["u"], d["R"], d["PK"], d["Gen"], d["m"]).all()
    assert not O.verify_vargen(d["u"], d["R"], np.roll(d["PK"], 1, 0), d["Gen"], d["m"]).any()
    assert not O.verify_vargen(d["u"], d["R"], d["PK"], np.roll(d["Gen"], 1, 0), d["m"]).any()


def test_projective_equality_semantics():
    """tests/keys.rs:33-59: 2G+7G and 4G+5G differ in (u, v, z) but are the same point."""
    L = O.lib()
    import ctypes

    class Ext(ctypes.Structure):
        _fields_ = [("l", ctypes.c_uint64 * 20)]

    g, a, b, c, t1, t2 = (Ext() for _ in range(6))
    L.oext_generator(ctypes.byref(g))

    def mul_small(out, k):
        s = (ctypes.c_uint8 * 32)(*M.le32(k))
        L.oext_mul(ctypes.byref(out), ctypes.byref(g), s)

    mul_small(t1, 2); mul_small(t2, 7); L.oext_add(ctypes.byref(a), ctypes.byref(t1), ctypes.byref(t2))
    mul_small(t1, 4); mul_small(t2, 5); L.oext_add(ctypes.byref(b), ctypes.byref(t1), ctypes.byref(t2))
    mul_small(t1, 4); mul_small(t2, 567758785); L.oext_add(ctypes.byref(c), ctypes.byref(t1), ctypes.byref(t2))
    L.oext_eq.restype = ctypes.c_int
    assert list(a.l[:12]) != list(b.l[:12])  # different projective representation
    assert L.oext_eq(ctypes.byref(a), ctypes.byref(b)) == 1
    assert L.oext_eq(ctypes.byref(a), ctypes.byref(c)) == 0


def test_tamper_classes_give_mixed_verdicts():
    d = O.keygen_sign_single(64, 5, nthreads=4)
    done = H.tamper(d, period=4)
    ok = O.verify_single(d["u"], d["R"], d["PK"], d["m"], nthreads=4)
    bad = {i for i, _ in done}
    for i in range(64):
        assert ok[i] == (0 if i in bad else 1), (i, dict(done).get(i))


def test_compress_decompress_roundtrip():
    import ctypes
    L = O.lib()
    L.ojub_decompress.restype = ctypes.c_int
    L.ojub_compress.restype = ctypes.c_int

    class Ext(ctypes.Structure):
        _fields_ = [("l", ctypes.c_uint64 * 20)]

    d = O.keygen_sign_single(8, 3)
    for i in range(8):
        p = H.to_int_point(d["PK"][i])
        comp = M.compress(p)
        e = Ext()
        buf = (ctypes.c_uint8 * 32)(*comp)
        assert L.ojub_decompress(ctypes.byref(e), buf) == 1
        out = (ctypes.c_uint8 * 32)()
        assert L.ojub_compress(out, ctypes.byref(e)) == 1
        assert bytes(out) == comp
    # not a curve point: v = 2 has no matching u?  find a v whose u^2 is a non-residue
    for v in range(2, 50):
        num = (v * v - 1) % M.Q
        den = (1 + M.D * v * v) % M.Q
        u2 = num * pow(den, -1, M.Q) % M.Q
        if pow(u2, (M.Q - 1) // 2, M.Q) == M.Q - 1:
            e = Ext()
            buf = (ctypes.c_uint8 * 32)(*M.le32(v))
            assert L.ojub_decompress(ctypes.byref(e), buf) == 0
            break


def test_golden_vectors_oracle_and_model():
    for rec in GOLDEN["single"]:
        u, R, PK, m = (unhex(rec[k]).reshape(1, -1) for k in ("u", "R", "PK", "m"))
        assert int(O.verify_single(u, R, PK, m)[0]) == rec["verdict"]
        assert bytes(O.challenge_single(R, m)[0]).hex() == rec["c"]
        assert M.compress(H.to_int_point(R[0])).hex() == rec["R_compressed"]
    for rec in GOLDEN["tampered_single"]:
        u, R, PK, m = (unhex(rec[k]).reshape(1, -1) for k in ("u", "R", "PK", "m"))
        assert int(O.verify_single(u, R, PK, m)[0]) == rec["verdict"]
    for rec in GOLDEN["double"]:
        a = {k: unhex(rec[k]).reshape(1, -1) for k in ("u", "R", "Rp", "PK", "PKp", "m")}
        assert int(O.verify_double(a["u"], a["R"], a["Rp"], a["PK"], a["PKp"], a["m"])[0]) == rec["verdict"]
        assert bytes(O.challenge_double(a["R"], a["Rp"], a["m"])[0]).hex() == rec["c"]
    for rec in GOLDEN["vargen"]:
        a = {k: unhex(rec[k]).reshape(1, -1) for k in ("u", "R", "PK", "Gen", "m")}
        assert int(O.verify_vargen(a["u"], a["R"], a["PK"], a["Gen"], a["m"])[0]) == rec["verdict"]
    for rec in GOLDEN["hash"]:
        msgs = [int(x, 16) for x in rec["inputs"]]
        assert hex(M.sponge_hash(msgs)) == rec["sponge"]
        assert hex(M.truncated_hash(msgs)) == rec["truncated"]
    assert [hex(x) for x in M.hades_permute([0, 1, 2, 3, 4])] == GOLDEN["hades_permute_0_1_2_3_4"]


def test_oracle_multithreaded_equals_single_thread():
    d = O.keygen_sign_single(100, 9, nthreads=3)
    d1 = O.keygen_sign_single(100, 9, nthreads=1)
    for k in d:
        assert np.array_equal(d[k], d1[k])
    H.tamper(d, period=5)
    assert np.array_equal(O.verify_single(d["u"], d["R"], d["PK"], d["m"], nthreads=7),
                          O.verify_single(d["u"], d["R"], d["PK"], d["m"], nthreads=1))


def test_stdrng_restatement_and_predicted_reference_vectors():
    """ChaCha core against the RFC 7539 block vector; the predicted reference outputs for the
    reference's own seeds regenerate byte-identically and verify (self-consistency only — the
    prediction itself is unverified until golden_gen.rs is run against the real crate)."""
    import refrng
    assert refrng.chacha20_rfc7539_selftest()
    P = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "predicted_reference.json")))
    sys_path = os.path.join(os.path.dirname(__file__), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_predicted", os.path.join(sys_path, "make_predicted.py"))
    mp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mp)
    assert mp.run(2321, 8) == P["seed_2321"]
    for rec in P["seed_2321"] + P["seed_0xbeef"]:
        assert rec["verdict"] == 1
        sig = unhex(rec["sig_bytes"]).reshape(1, 64)
        pk = unhex(rec["pk_bytes"]).reshape(1, 32)
        assert int(O.verify_single_wire(sig, pk, unhex(rec["m"]).reshape(1, 32))[0]) == 1


def _check_reference_records(recs, verify_wire, verify_plain, decompress):
    """shared by the CPU (oracle) and GPU (engine) forms of the hand-off test"""
    n_sig = 0
    for r in recs:
        if r["kind"] == "sponge_hash_1_2_3_le":
            assert M.le32(M.sponge_hash([1, 2, 3])).hex() == r["hex"], "Poseidon sponge differs from the crate"
        elif r["kind"] == "truncated_hash_1_2_3_le":
            assert M.le32(M.truncated_hash([1, 2, 3])).hex() == r["hex"], "250-bit truncation differs"
        elif r["kind"] == "sig":
            n_sig += 1
            u, m = unhex(r["u"]).reshape(1, 32), unhex(r["m"]).reshape(1, 32)
            R, PK = unhex(r["R"]).reshape(1, 64), unhex(r["PK"]).reshape(1, 64)
            assert int(verify_plain(u, R, PK, m)[0]) == int(r["verdict"]), ("verdict", r["i"])
            sig, pk = unhex(r["sig_bytes"]).reshape(1, 64), unhex(r["pk_bytes"]).reshape(1, 32)
            assert int(verify_wire(sig, pk, m)[0]) == int(r["verdict"]), ("wire verdict", r["i"])
            # the crate's own PK = sk * G and serialisation
            sk = M.from_le(unhex(r["sk"]))
            assert M.point_bytes(M.pmul(M.GEN, sk)).hex() == r["PK"], ("PK = sk*G", r["i"])
            assert M.compress(H.to_int_point(PK[0])).hex() == r["pk_bytes"]
            assert r["sig_bytes"] == r["u"] + M.compress(H.to_int_point(R[0])).hex()
        elif r["kind"] == "from_bytes":
            out, ok = decompress(unhex(r["enc"]).reshape(1, 32))
            assert bool(ok[0]) == r["ok"], ("from_bytes accept/reject", r["i"])
            if r["ok"]:
                assert bytes(out[0]).hex() == r["u"] + r["v"], ("from_bytes value", r["i"])
    return n_sig


def test_reference_fixtures_pin_the_oracle():
    """PARITY HAND-OFF: fixtures dumped from the real dusk-schnorr (tests/reference_fixtures.py,
    rust/dusk-schnorr-gpu/src/bin/golden_gen.rs), when someone has dropped them into
    tests/golden/, must agree with the oracle: raw sponge, truncation, sign / key bytes, verdicts,
    decompression edge cases.  Skipped (parity stays "unpinned") while there is none."""
    import reference_fixtures as RF
    recs = RF.load()
    if not recs:
        pytest.skip("no tests/golden/reference_* fixture present: parity unpinned (DESIGN.md §2)")
    n = _check_reference_records(recs, O.verify_single_wire, O.verify_single, O.decompress)
    assert n >= 1


def test_reference_fixture_parser_on_a_synthetic_file(tmp_path, monkeypatch):
    """the hand-off path itself is exercised: a file in golden_gen.rs's format built from the
    PREDICTED vectors parses and passes; one corrupted challenge-dependent byte fails."""
    import reference_fixtures as RF
    P = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "predicted_reference.json")))
    lines = ["sponge_hash_1_2_3_le " + M.le32(M.sponge_hash([1, 2, 3])).hex(),
             "truncated_hash_1_2_3_le " + M.le32(M.truncated_hash([1, 2, 3])).hex()]
    for rec in P["seed_2321"][:3]:
        lines.append("sig %d sk %s m %s u %s R %s PK %s sig_bytes %s pk_bytes %s verdict true"
                     % (rec["i"], rec["sk"], rec["m"], rec["u"], rec["R"], rec["PK"], rec["sig_bytes"],
                        rec["pk_bytes"]))
    one = M.le32(1).hex()
    lines.append("from_bytes 0 %s ok u %s v %s" % (one, M.le32(0).hex(), one))
    (tmp_path / "reference_synthetic.txt").write_text("\n".join(lines) + "\n")
    monkeypatch.setattr(RF, "GOLDEN_DIR", str(tmp_path))
    recs = RF.load()
    assert len(recs) == 6
    assert _check_reference_records(recs, O.verify_single_wire, O.verify_single, O.decompress) == 3
    bad = [dict(r) for r in recs]
    u = bytearray(unhex(bad[2]["u"]).tobytes())
    u[0] ^= 1
    bad[2]["u"] = bytes(u).hex()
    with pytest.raises(AssertionError):
        _check_reference_records(bad, O.verify_single_wire, O.verify_single, O.decompress)
