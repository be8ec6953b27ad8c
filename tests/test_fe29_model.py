"""The lazy-reduction contracts of fe29.h / jubjub29.h / hades29.h, checked on a bit-exact Python
model (tests/fe29_model.py) whose every 32/64-bit overflow condition is an assertion: random
inputs, adversarial all-ones limbs, and long chains of the exact formula sequences the kernels run."""
import random

import fe29_model as F
import pymodel as M

rnd = random.Random(1234)


def rand_fe():
    return F.to_mont_int(rnd.randrange(F.Q))


def test_mul_sqr_add_sub_values():
    for _ in range(300):
        x, y = rnd.randrange(F.Q), rnd.randrange(F.Q)
        a, b = F.to_mont_int(x), F.to_mont_int(y)
        assert F.val(F.from_mont(F.mul(a, b))) == x * y % F.Q
        assert F.val(F.from_mont(F.sqr(a))) == x * x % F.Q
        assert F.val(F.from_mont(F.mul(F.add(a, b), F.sub(a, b, 2)))) == (x + y) * (x - y) % F.Q
        assert F.equal(F.sub(a, b, 4), F.to_mont_int((x - y) % F.Q))
    for x in (0, 1, F.Q - 1, (1 << 254), F.M29, 1 << 29):
        a = F.to_mont(F.from_int(x))
        assert F.val(F.from_mont(a)) == x % F.Q


def test_worst_case_limbs_do_not_overflow_columns():
    """limbs at the contract ceiling: a < 2^30 (sum of two normalised), b < 1.5*2^30 (d + c)"""
    a = [(1 << 30) - 2] * 8 + [(1 << 26)]
    b = [3 * (1 << 29)] * 8 + [(1 << 26)]
    F.mul(a, b)
    F.mul(b, a)
    F.sqr(a)
    assert F.stats["max_col"] < (1 << 64)


def test_group_law_chain_matches_affine_model_and_keeps_bounds():
    """the exact op sequence of var_base_mul / fixed_base_accumulate for a random scalar"""
    for trial in range(2):
        k = rnd.randrange(1 << 250)
        P = M.pmul(M.GEN, rnd.randrange(1, M.R_ORDER))
        pu, pv = F.to_mont_int(P[0]), F.to_mont_int(P[1])
        p = F.ext_from_affine(pu, pv)
        n1 = F.ext_to_niels(p)
        tbl = [{"vpu": list(F.ONE), "vmu": list(F.ONE), "z": list(F.ONE), "t2d": [0] * 9}, n1]
        cur = p
        for _ in range(2, 16):
            cur = F.ext_add_niels(cur, n1)
            tbl.append(F.ext_to_niels(cur))
        acc = F.ext_identity()
        for d in range(62, -1, -1):
            for _ in range(4):
                acc = F.ext_double(acc)
            acc = F.ext_add_niels(acc, tbl[(k >> (4 * d)) & 15])
        assert F.affine_of(acc) == M.pmul(P, k)
        # continue with mixed additions of affine-niels points (fixed-base stage)
        tot = M.pmul(P, k)
        for w in range(6):
            Qp = M.pmul(M.GEN, rnd.randrange(1, M.R_ORDER))
            u, v = Qp
            an = {"vpu": F.to_mont_int((v + u) % F.Q), "vmu": F.to_mont_int((v - u) % F.Q),
                  "t2d": F.to_mont_int(2 * M.D * u * v % F.Q)}
            acc = F.ext_add_aniels(acc, an)
            tot = M.padd(tot, Qp)
        assert F.affine_of(acc) == tot
        ru, rv = F.to_mont_int(tot[0]), F.to_mont_int(tot[1])
        assert F.equal(acc["u"], F.mul(ru, acc["z"])) and F.equal(acc["v"], F.mul(rv, acc["z"]))


def test_chain_start_and_mixed_table_build_match_the_affine_model():
    """ext_from_niels(n) == O + n, and the table build by mixed additions (P affine) gives i*P"""
    for _ in range(3):
        P = M.pmul(M.GEN, rnd.randrange(1, M.R_ORDER))
        p = F.ext_from_affine(F.to_mont_int(P[0]), F.to_mont_int(P[1]))
        n1 = F.ext_to_niels(p)
        a1 = {k: n1[k] for k in ("vpu", "vmu", "t2d")}
        cur = p
        for i in range(2, 9):
            cur = F.ext_add_aniels(cur, a1)
            assert F.affine_of(cur) == M.pmul(P, i)
            assert F.affine_of(F.ext_from_niels(F.ext_to_niels(cur))) == M.pmul(P, i)
            # and the result feeds the next operations like any other accumulator
            nxt = F.ext_add_niels(F.ext_double(F.ext_from_niels(F.ext_to_niels(cur))), n1)
            assert F.affine_of(nxt) == M.pmul(P, 2 * i + 1)
    ident = {"vpu": list(F.ONE), "vmu": list(F.ONE), "z": list(F.ONE), "t2d": [0] * 9}
    assert F.affine_of(F.ext_from_niels(ident)) == (0, 1)


def test_shared_sum_and_difference_of_the_joint_table_build():
    """ext_add_sub_aniels_t: (p + n, p - n) from one shared addition, for affine and projective p"""
    for _ in range(4):
        P = M.pmul(M.GEN, rnd.randrange(1, M.R_ORDER))
        R = M.pmul(M.GEN, rnd.randrange(1, M.R_ORDER))
        p = F.ext_from_affine(F.to_mont_int(P[0]), F.to_mont_int(P[1]))
        r = F.ext_from_affine(F.to_mont_int(R[0]), F.to_mont_int(R[1]))
        nr = F.ext_to_niels(r)
        ar = {k: nr[k] for k in ("vpu", "vmu", "t2d")}
        assert F.affine_of(F.ext_double_affine(p["u"], p["v"])) == M.pmul(P, 2)
        for base, want in ((p, P), (F.ext_double_affine(p["u"], p["v"]), M.pmul(P, 2)),
                           (F.ext_add_niels(F.ext_double(F.ext_double(p)), nr), M.padd(M.pmul(P, 4), R))):
            s_, d_ = F.ext_add_sub_aniels(base, ar)
            assert F.affine_of(s_) == M.padd(want, R)
            assert F.affine_of(d_) == M.padd(want, M.pneg(R))
            # both feed the next operations like any other accumulator
            assert F.affine_of(F.ext_double(d_)) == M.pmul(M.padd(want, M.pneg(R)), 2)
            assert F.affine_of(F.ext_from_niels(F.ext_to_niels(d_))) == M.padd(want, M.pneg(R))


def test_identity_decided_inside_the_last_addition():
    """ext_add_aniels_is_identity(p, n) == [p + n == O], for p = -n (true), p = -n + torsion and
    random p (false), affine and projective p"""
    import test_halfgcd as TH
    t8 = TH.order8_point()
    for trial in range(6):
        N = M.pmul(M.GEN, rnd.randrange(1, M.R_ORDER))
        n_ext = F.ext_from_affine(F.to_mont_int(N[0]), F.to_mont_int(N[1]))
        nn = F.ext_to_niels(n_ext)
        an = {k: nn[k] for k in ("vpu", "vmu", "t2d")}
        for P in (M.pneg(N), M.padd(M.pneg(N), M.pmul(t8, 1 + trial)), M.pmul(M.GEN, rnd.randrange(1, M.R_ORDER)),
                  M.IDENTITY, N):
            p = F.ext_from_affine(F.to_mont_int(P[0]), F.to_mont_int(P[1]))
            # a projective representative: p = (2P - P) through the formulas
            p2 = F.ext_add_niels(F.ext_double(p), F.ext_to_niels(F.ext_from_affine(
                F.to_mont_int(M.pneg(P)[0]), F.to_mont_int(M.pneg(P)[1]))))
            want = M.padd(P, N) == M.IDENTITY
            assert F.ext_add_aniels_is_identity(p, an) == want
            assert F.ext_add_aniels_is_identity(p2, an) == want
            assert (F.affine_of(F.ext_add_aniels(p2, an)) == (0, 1)) == want


def test_identity_and_torsion_through_the_formulas():
    ident = F.ext_identity()
    d = F.ext_double(ident)
    assert F.affine_of(d) == (0, 1)
    o2 = F.ext_from_affine(F.to_mont_int(0), F.to_mont_int(F.Q - 1))
    assert F.affine_of(F.ext_double(o2)) == (0, 1)
    s = F.ext_add_niels(o2, F.ext_to_niels(o2))
    assert F.affine_of(s) == (0, 1)
    s = F.ext_add_niels(ident, F.ext_to_niels(ident))
    assert F.affine_of(s) == (0, 1)
