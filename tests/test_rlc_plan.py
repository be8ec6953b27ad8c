"""Geometry of the batch fast accept (schnorr_amd/csrc/rlc.h: rlc_plan) — CPU, through
the C ABI (dsv_rlc_plan_info needs no GPU).  The kernels of k_rlc.hip index their buffers by these
numbers alone; what they assume is asserted here for every scheme, window width and a sweep of sizes."""
import pytest

from schnorr_amd import _lib, engine as E

R_ORDER = 0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7
SIZES = [1, 2, 63, 64, 65, 1000, 4095, 4096, (1 << 14) - 1, 1 << 14, (1 << 17) - 1, 1 << 17, (1 << 19) - 1, 1 << 19,
         (1 << 20) + 12345, 1 << 22]


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_plan_invariants(scheme):
    lpts, spts, fixed = {"single": (1, 1, 1), "double": (2, 2, 2), "vargen": (2, 1, 0)}[scheme]
    for n in SIZES:
        for bits in (0, 4, 6, 8, 12, 14, 16):
            p = E.rlc_plan_info(scheme, n, bits)
            c = p["c"]
            assert c in (4, 6, 8, 12, 14, 16) and (bits == 0 or c == bits)
            assert p["half"] * 2 == c
            assert (p["lpts"], p["spts"], p["fixed"]) == (lpts, spts, fixed)
            # the windows cover the scalars: keys 252 bits (+ the multiples of r: 256 at most), nonce weights >= 128
            assert p["wpk"] * c >= 252 and (p["wpk"] - 1) * c < 252 and p["wpk"] * c <= 256
            assert p["wr"] * c >= 128 and (p["wr"] - 1) * c < 128 and p["wr"] * c <= 160   # five keystream words
            assert p["windows"] == p["wpk"] + p["wr"] and p["windows"] < 128
            # e + k r < 2^(wpk c) for every k < kmul, and kmul is the largest such count
            assert p["kmul"] == (1 << (p["wpk"] * c)) // R_ORDER >= 1
            assert (R_ORDER - 1) + (p["kmul"] - 1) * R_ORDER < 1 << (p["wpk"] * c) <= 1 << 256
            # sort keys: every bucket number and the "digit 0" key (= buckets) fit key_bits
            assert p["buckets"] == p["windows"] << c
            assert p["buckets"] < 1 << p["key_bits"] <= 1 << 31
            assert p["entries"] == n * (p["wpk"] * lpts + p["wr"] * spts) < 1 << 32
            assert (lpts + spts) * n < 1 << 32
            # row / column and bit-sum chains: whole segments, none longer than 16 (or the whole side when small)
            side = 1 << p["half"]
            assert side % p["nseg"] == 0 and (side // 2) % p["nseg2"] == 0
            assert side // p["nseg"] <= 16 or p["nseg"] == 1 and side <= 16
            assert side // 2 // p["nseg2"] <= 16 or p["nseg2"] == 1
            # scratch areas hold every stage that writes them (k_rlc.hip: launch_rlc)
            lanes = p["windows"] * c
            assert p["tmp0"] >= max(p["windows"] * 2 * side * p["nseg"], p["windows"] * 2 * p["half"] * p["nseg2"], lanes + 1)
            assert p["tmp1"] >= max(p["windows"] * 2 * side, lanes)
            assert E.rlc_workspace_bytes(n, bits) > 0


def test_default_window_widths_follow_the_batch_size():
    assert [E.rlc_plan_info("single", n)["c"] for n in (100, (1 << 14) - 1, 1 << 14, 1 << 17, 1 << 19, 1 << 22)] == \
        [8, 8, 12, 14, 16, 16]


def test_plan_argument_checks():
    import ctypes
    L = _lib.load()
    out = (ctypes.c_uint64 * 16)()
    assert L.dsv_rlc_plan_info(ctypes.c_int(3), ctypes.c_size_t(10), ctypes.c_int(0), out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t(0), ctypes.c_int(0), out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t(10), ctypes.c_int(10), out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t((1 << 22) + 1), ctypes.c_int(0), out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t(10), ctypes.c_int(0), None) == -2
    assert int(L.dsv_rlc_workspace_bytes(ctypes.c_size_t(10), ctypes.c_int(10))) == 0
