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
            assert p["buckets"] == p["windows"] << c
            assert p["groups"] == 1 and p["sub"] == n
            _check_partition(p, n, lpts, spts)
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


def _check_partition(p, sub, lpts, spts):
    """the two-pass partition of the digits (k_rlc_part1 / k_rlc_part2): what the kernels index by"""
    c = p["c"]
    # a 16-bit digit per (window, point slot, item); the low bits sorted through LDS in 256 counters
    assert c <= 16 and p["fine_bits"] == min(c, 8) and p["fine_bits"] + p["coarse_bits"] == c
    assert p["rows"] == p["wpk"] * lpts + p["wr"] * spts
    assert p["entries"] == sub * p["rows"] < 1 << 32
    # rows start 16-byte aligned (uint4 loads of eight digits) and hold every item of the sub-group
    assert p["row_stride"] % 8 == 0 and sub <= p["row_stride"] < sub + 8
    # k_rlc_part1 reserves with one LDS counter per bin of a window: at most 256 bins per window
    assert p["bins"] == p["windows"] << p["coarse_bits"] and (1 << p["coarse_bits"]) <= 256
    # (low digit bits << 24 | point index): point indices of a sub-group fit 24 bits
    assert (lpts + spts) * sub <= 1 << 24
    # a bin takes its mean load + 8 standard deviations, or everything a window can hold.  Mean: uniform digits,
    # except that the keys' top window is uniform over kmul r / 2^(wpk c) = 0.905 / 0.962 of its digits only
    most = max(lpts, spts) * sub
    mean = most / (1 << p["coarse_bits"]) * (1 << (p["wpk"] * c)) / (p["kmul"] * R_ORDER)
    assert p["bin_cap"] % 64 == 0 and p["bin_cap"] >= min(most, mean + 8 * mean ** 0.5)
    assert p["bins"] * p["bin_cap"] < 1 << 32


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_sub_group_plans(scheme):
    """a group cut into sub-groups (rlc.h: RlcPlan.groups): whole sub-batches of the per-signature path, the
    last one ragged, never an empty one; window bits follow the SUB-group's size"""
    lpts, spts, _ = {"single": (1, 1, 1), "double": (2, 2, 2), "vargen": (2, 1, 0)}[scheme]
    for n in (2, 1000, (1 << 17) + 5, 1 << 18, (1 << 20) - 3, 1 << 20, (1 << 20) + 12345, 1 << 22):
        for groups in (2, 3, 4, 8, 16):
            p = E.rlc_plan_info(scheme, n, 0, groups)
            G, sub = p["groups"], p["sub"]
            assert 1 <= G <= groups and (G - 1) * sub < n <= G * sub
            if n >= 1 << 17:
                assert sub % (1 << 16) == 0 or G == 1     # no launch of the fallback straddles two sub-groups
            assert p["c"] == E.rlc_plan_info(scheme, sub)["c"]
            _check_partition(p, sub, lpts, spts)
            assert p["bytes"] <= E.rlc_workspace_bytes(n)
            p8 = E.rlc_plan_info(scheme, n, 8, groups)
            assert p8["c"] == 8 and p8["bytes"] <= E.rlc_workspace_bytes(n, 8)


def test_workspace_size_never_drops_with_the_batch_size():
    """dsv_verify_mixed_rlc_dev sizes ONE fast-accept workspace for n items and then runs
    aggregates over ns and nd <= n items of either kind — the size for n must cover every count up to n,
    also just above 2^22 items, where a call's own groups shrink to n / 2."""
    sizes = [1, 1000, 1 << 17, (1 << 19) - 1, 1 << 19, 1 << 20, (1 << 22) - 1, 1 << 22, (1 << 22) + 2, 5_000_000,
             (1 << 23) - 7, 1 << 23, (1 << 23) + 1, 1 << 24]
    prev = 0
    for n in sizes:
        b = E.rlc_workspace_bytes(n)
        assert b >= prev, (n, b, prev)
        prev = b
    # the mixed entry point's share for the aggregates is that size
    for n in ((1 << 22) + 2, 5_000_000, 1 << 23):
        whole = E.mixed_rlc_workspace_bytes(n)
        for k in (n // 2, n - 1, 1 << 22, 4_000_000):
            if k <= n:
                assert E.rlc_workspace_bytes(k) <= E.rlc_workspace_bytes(n) <= whole


def test_default_window_widths_follow_the_batch_size():
    assert [E.rlc_plan_info("single", n)["c"] for n in (100, (1 << 14) - 1, 1 << 14, 1 << 17, 1 << 19, 1 << 20, 1 << 22)] == \
        [8, 8, 12, 14, 14, 16, 16]


def test_plan_argument_checks():
    import ctypes
    L = _lib.load()
    out = (ctypes.c_uint64 * 24)()
    one = ctypes.c_int(1)
    assert L.dsv_rlc_plan_info(ctypes.c_int(3), ctypes.c_size_t(10), ctypes.c_int(0), one, out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t(0), ctypes.c_int(0), one, out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t(10), ctypes.c_int(10), one, out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t((1 << 22) + 1), ctypes.c_int(0), one, out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t(10), ctypes.c_int(0), ctypes.c_int(17), out) == -2
    assert L.dsv_rlc_plan_info(ctypes.c_int(0), ctypes.c_size_t(10), ctypes.c_int(0), one, None) == -2
    assert int(L.dsv_rlc_workspace_bytes(ctypes.c_size_t(10), ctypes.c_int(10))) == 0
