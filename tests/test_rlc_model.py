"""The batch fast accept (schnorr_amd/csrc/k_rlc.hip) as a Python-integer model — CPU.

What the kernels compute, step for step (prep -> buckets -> row / column sums -> per-bit subset sums
S_p -> "r * S_p == O for all p" and "sum 2^p S_p + (sum z_i u_i) G == O"), on the affine group law of
tests/pymodel.py, for the equation of `PublicKey::verify` (/root/reference/src/keys/public.rs:121-130).
It pins the three facts the design rests on, independently of the GPU:
  * the row / column sums of a window's bucket matrix give the per-bit subset sums, and
    sum_p 2^p S_p is the multi-scalar sum (so no running-sum reduction is needed);
  * adding a multiple of r to a key's scalar changes nothing for points of the prime-order subgroup;
  * a plain weighted sum of the equations ACCEPTS two order-2 defects whenever z_a + z_b is even, while
    the subgroup test on the S_p rejects every batch that holds a point with a small-order component —
    the reason the aggregate is exact on the reference's cofactorless equation.
"""
import random

import pymodel as M
import test_halfgcd as TH

R = M.R_ORDER


def _sign(rnd, torsion_R=None, torsion_PK=None):
    sk, m, rr = rnd.randrange(1, R), rnd.randrange(M.Q), rnd.randrange(1, R)
    pk = M.pmul(M.GEN, sk)
    Rp = M.pmul(M.GEN, rr)
    if torsion_PK:
        pk = M.padd(pk, torsion_PK)
    if torsion_R:
        Rp = M.padd(Rp, torsion_R)
    c = M.challenge(Rp, m)
    return {"u": (rr - c * sk) % R, "R": Rp, "PK": pk, "m": m, "c": c}


def _aggregate(items, rnd, c_bits=4):
    """-> (subgroup_ok, sum_ok, naive_ok): the kernels' two tests and the plain weighted sum"""
    half = c_bits // 2
    wpk, wr = -(-252 // c_bits), -(-128 // c_bits)
    kmul = (1 << (wpk * c_bits)) // R
    side = 1 << half
    # scalars (k_rlc_prep): z on -R, z c + k r on PK, z u summed for G
    long_pts, short_pts, fsum, naive = [], [], 0, M.IDENTITY
    for it in items:
        z = rnd.getrandbits(wr * c_bits)
        e = z * it["c"] % R + rnd.randrange(kmul) * R
        long_pts.append((it["PK"], e))
        short_pts.append((M.pneg(it["R"]), z))
        fsum = (fsum + z * it["u"]) % R
        d = M.padd(M.padd(M.pmul(M.GEN, it["u"]), M.pmul(it["PK"], it["c"])), M.pneg(it["R"]))
        naive = M.padd(naive, M.pmul(d, z))
    S = []   # (position in its scalar, subset sum) for every bit of every window
    total = M.IDENTITY
    for pts, windows in ((long_pts, wpk), (short_pts, wr)):
        direct = M.IDENTITY
        for P, s in pts:
            direct = M.padd(direct, M.pmul(P, s))
        acc = M.IDENTITY
        for w in range(windows):
            buckets = [[M.IDENTITY] * side for _ in range(side)]          # [high half][low half]
            for P, s in pts:
                d = (s >> (c_bits * w)) & ((1 << c_bits) - 1)
                if d:                                                      # digit 0 enters no sum
                    buckets[d >> half][d & (side - 1)] = M.padd(buckets[d >> half][d & (side - 1)], P)
            rows = [M.IDENTITY] * side
            cols = [M.IDENTITY] * side
            for h in range(side):
                for lo in range(side):
                    rows[h] = M.padd(rows[h], buckets[h][lo])
                    cols[lo] = M.padd(cols[lo], buckets[h][lo])
            for kind, lines in ((0, cols), (1, rows)):                     # low bits from the columns, high from the rows
                for j in range(half):
                    sp = M.IDENTITY
                    for idx in range(side):
                        if (idx >> j) & 1:
                            sp = M.padd(sp, lines[idx])
                    pos = c_bits * w + kind * half + j
                    S.append(sp)
                    acc = M.padd(acc, M.pmul(sp, 1 << pos))
        assert acc == direct                 # sum_p 2^p S_p IS the multi-scalar sum
        total = M.padd(total, acc)
    subgroup_ok = all(M.pmul(sp, R) == M.IDENTITY for sp in S)
    sum_ok = M.padd(total, M.pmul(M.GEN, fsum)) == M.IDENTITY
    return subgroup_ok, sum_ok, naive == M.IDENTITY


def test_valid_batch_is_accepted_and_a_wrong_signature_is_not():
    rnd = random.Random(11)
    items = [_sign(rnd) for _ in range(6)]
    assert _aggregate(items, rnd) == (True, True, True)
    items[2]["u"] = (items[2]["u"] + 1) % R
    sub, total, naive = _aggregate(items, rnd)
    assert sub and not total and not naive


def test_random_multiples_of_r_do_not_change_a_prime_order_point():
    rnd = random.Random(12)
    P = M.pmul(M.GEN, rnd.randrange(1, R))
    e = rnd.randrange(R)
    assert M.pmul(P, e + 16 * R) == M.pmul(P, e)
    t8 = TH.order8_point()
    assert M.pmul(M.padd(P, t8), e + R) != M.pmul(M.padd(P, t8), e)      # ... and do for any other point


def test_cancelling_order_two_defects_pass_the_plain_sum_but_not_the_subgroup_test():
    rnd = random.Random(13)
    t2 = M.pmul(TH.order8_point(), 4)
    assert t2 == (0, M.Q - 1)
    items = [_sign(rnd) for _ in range(3)] + [_sign(rnd, torsion_R=t2), _sign(rnd, torsion_R=t2)]
    for it in items[3:]:                                                   # each one alone is invalid
        assert not M.verify_single(it["u"], it["R"], it["PK"], it["m"])
    seen_naive_accept = False
    for _ in range(6):
        sub, total, naive = _aggregate(items, rnd)
        assert not sub                                                     # never decided by the aggregate
        seen_naive_accept |= naive
    assert seen_naive_accept                                               # the plain sum accepts for even z_a + z_b


def test_a_valid_signature_with_cancelling_torsion_still_fails_the_subgroup_test():
    """key and nonce point with order-8 components that cancel in the reference's equation (c k1 = k2 mod 8):
    the item is VALID, the batch all valid — and the aggregate must not decide it (the per-signature kernels do)"""
    rnd = random.Random(14)
    t8 = TH.order8_point()
    while True:
        k1, k2 = rnd.randrange(1, 8), rnd.randrange(8)
        it = _sign(rnd, torsion_R=M.pmul(t8, k2) if k2 else None, torsion_PK=M.pmul(t8, k1))
        if (it["c"] * k1 - k2) % 8 == 0:
            break
    assert M.verify_single(it["u"], it["R"], it["PK"], it["m"])
    sub, total, naive = _aggregate([_sign(rnd), it, _sign(rnd)], rnd)
    assert not sub
