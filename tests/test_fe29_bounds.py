"""Worst-case (all-inputs) overflow proof of the lazy-reduction bookkeeping in fe29.h /
jubjub29.h — see tests/fe29_bounds.py.  CPU only."""
import pytest

import fe29_bounds as FB


def test_group_law_bounds_reach_a_fixpoint_without_overflow():
    inv = FB.prove_group_law()
    acc = inv["acc"]
    # what the comments in jubjub29.h promise: multiplication outputs are "N"
    for k in ("u", "v", "z"):
        assert acc[k].l[0] <= FB.M29 + 1           # limb 0 takes the accumulator's missing 1 back
        assert all(x <= FB.M29 for x in acc[k].l[1:8])
        assert acc[k].v < 2 * FB.Q
    assert inv["rounds"] < 10


def test_prover_rejects_a_broken_variant():
    """Sanity of the prover itself: dropping BOTH carry passes of the doubling must be refused."""
    def bad_double(p):
        uu, vv = FB.sqr(p["u"]), FB.sqr(p["v"])
        zz2 = FB.dbl(FB.sqr(p["z"]))
        cu = FB.dbl(FB.mul(p["u"], p["v"]))
        vpu = FB.add(vv, uu)
        vmu = FB.sub_raw(vv, uu, 2)
        ct = FB.sub_raw(zz2, vmu, 4)
        return {"u": FB.mul(cu, ct), "v": FB.mul(vpu, vmu), "z": FB.mul(vmu, ct), "t1": cu, "t2": vpu}

    n = FB.mul(FB.canonical(), FB.canonical())
    with pytest.raises(FB.OverflowError_):
        bad_double({"u": n, "v": n, "z": n, "t1": n, "t2": n})


def test_hades_permutation_cannot_overflow_with_the_shipped_constants():
    """S-boxes on the VALU between rows that come back from the matrix cores (hades29.h)"""
    out = FB.prove_hades()
    assert out["hash3"].v < 3 * FB.Q and out["hash5"].v < 3 * FB.Q


def test_point_decompression_field_code_cannot_overflow():
    FB.prove_decompress()


def test_normalisation_and_limb_conversion_field_code_cannot_overflow():
    """r04: k_scalars_from_mont and k_normalize_uvz with the Euclidean inversion (inv29.h)"""
    out = FB.prove_normalize_and_limb_conversion()
    assert out["quotient"].v < 3 * FB.Q


def test_batch_fast_accept_field_code_cannot_overflow():
    """r05, k_rlc.hip: the curve test, the affine-niels form of decoded points and the four-waves-per-point
    operations (running point with tt = t1 * t2) stay inside the group law's proven invariants"""
    out = FB.prove_fast_accept()
    assert out["rounds"] < 12 and out["acc"]["u"].v < 2 * FB.Q
