"""CPU tests of the oracle's restatement of verify on the reference's IN-MEMORY representation
(four u64 Montgomery limbs per element, R = 2^256: /root/reference/Cargo.toml:25-26,
src/signatures.rs:58-61, src/keys/public.rs:59) — the checker of the dsv_verify_*_mont entry
points (tests/test_gpu_r04.py)."""
import numpy as np
import pytest

import mont_cases as C
import oracle_lib as O
import pymodel as M


def test_limb_conversions_match_python_integers():
    rng = np.random.default_rng(1)
    for fr, mod in ((False, M.Q), (True, M.R_ORDER)):
        vals = [0, 1, 2, mod - 1, mod - 2, (1 << 250) - 1] + [int.from_bytes(rng.bytes(40), "little") % mod
                                                              for _ in range(200)]
        x = np.array([np.frombuffer(M.le32(v), np.uint8) for v in vals])
        limbs = O.to_mont(x, fr=fr)
        assert np.array_equal(limbs, C.to_limbs_py(x, mod))
        back, good = O.from_mont(limbs, fr=fr)
        assert good and np.array_equal(back, x)
        # limbs that are not below the modulus are reported (the Rust types cannot hold them)
        bad = limbs.copy()
        bad[3] = np.frombuffer(M.le32(mod), np.uint8)
        assert not O.from_mont(bad, fr=fr)[1]
    # R = 2^256 mod q, the limbs of BlsScalar::one()
    one = O.to_mont(np.frombuffer(M.le32(1), np.uint8).reshape(1, 32))
    assert one.view("<u8").tolist() == [[0x00000001fffffffe, 0x5884b7fa00034802, 0x998c4fefecbc4ff5, 0x1824b159acc5056f]]
    one_r = O.to_mont(np.frombuffer(M.le32(1), np.uint8).reshape(1, 32), fr=True)
    assert one_r.view("<u8").tolist() == [[0x25f80bb3b99607d9, 0xf315d62f66b6e750, 0x932514eeeb8814f4, 0x09a6fc6f479155c6]]


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_verify_on_limbs_equals_verify_on_canonical_bytes(scheme):
    """mont_case itself asserts oracle_verify_*_mont == oracle_verify_*_ext on the same values, with
    z = 0 / limbs >= q / limbs >= r items forced to 0; here: the batch is non-trivial and the record
    views a binding would pass hold the same bytes as the dense columns."""
    n = 96
    cols, want = C.mont_case(scheme, n, {"single": 21, "double": 22, "vargen": 23}[scheme])
    assert 0 < want.sum() < n
    sigs, pks, msgs, views = C.as_records(scheme, cols)
    assert sigs.dtype.itemsize == {"single": 192, "double": 352, "vargen": 192}[scheme]
    assert pks.dtype.itemsize == {"single": 160, "double": 320, "vargen": 320}[scheme]
    for v, c in zip(views, cols):
        assert v.shape == c.shape and np.array_equal(v, c)
    assert views[1].strides[0] == sigs.dtype.itemsize and not views[1].flags["C_CONTIGUOUS"]
    # a common factor of (u, v, z) does not change the verdict: re-scale one point's limbs
    again = [c.copy() for c in cols]
    k = int(np.flatnonzero(want)[0])
    pt = again[1][k]
    for part in range(3):
        v = M.from_le(pt[32 * part:32 * part + 32]) * 12345 % M.Q
        pt[32 * part:32 * part + 32] = np.frombuffer(M.le32(v), np.uint8)
    assert np.array_equal(getattr(O, "verify_%s_mont" % scheme)(*again), want)
