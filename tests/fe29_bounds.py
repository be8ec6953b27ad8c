"""Worst-case bound prover for schnorr_amd/csrc/fe29.h + jubjub29.h.  TEST INFRA.

tests/fe29_model.py executes the limb code on concrete values and asserts that nothing overflows
on THOSE values.  This module is the complement: an abstract interpretation in which a field
element is (per-limb upper bounds, value upper bound), every limb and value is assumed to be
anything in [0, bound], and each device operation maps bounds to bounds while asserting that no
32-bit limb, 64-bit column or biased subtraction can overflow / underflow for ANY input inside
the bounds.  All operations are monotone in their inputs, so evaluating them at the upper bounds
gives valid upper bounds of the results.

The point-level functions mirror jubjub29.h line by line.  `prove_group_law()` iterates them from
the widest inputs the kernels can produce until the bounds stop growing (a fixpoint) — i.e. it
proves the lazy-reduction bookkeeping of an arbitrarily long double/add chain.
"""
import os
import re

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
NL, LB = 9, 29
M29 = (1 << LB) - 1
RBITS = NL * LB
U32, U64 = 1 << 32, 1 << 64

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schnorr_amd", "csrc",
                    "dsv_constants.h")


def _load(name):
    text = open(_HDR).read()
    m = re.search(r"#define %s \{([^}]*)\}" % name, text)
    return [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]


Q29 = _load("DSV_Q29")
BIAS = {2: _load("DSV_BIAS2"), 4: _load("DSV_BIAS4"), 8: _load("DSV_BIAS8"),
        "4w": _load("DSV_BIAS4W")}
BIAS_MULT = {2: 2, 4: 4, 8: 8, "4w": 4}
for _k, _b in BIAS.items():  # the redundant-limb biases are exact multiples of q
    assert sum(x << (LB * i) for i, x in enumerate(_b)) == BIAS_MULT[_k] * Q


class OverflowError_(AssertionError):
    pass


def _check(cond, msg):
    if not cond:
        raise OverflowError_(msg)


class B:
    """Bounds of one field element: l[i] = max of limb i, v = max of the integer value."""

    __slots__ = ("l", "v")

    def __init__(self, limbs, v=None):
        self.l = list(limbs)
        full = sum(x << (LB * i) for i, x in enumerate(self.l))
        self.v = full if v is None else min(v, full)
        # the top limb can never exceed what the value allows (lower limbs are >= 0)
        self.l[NL - 1] = min(self.l[NL - 1], self.v >> (LB * (NL - 1)))

    def __repr__(self):
        return "B(limbs<=%s, v<=%.3f q)" % ([hex(x) for x in self.l], self.v / Q)


def join(a, b):
    return B([max(x, y) for x, y in zip(a.l, b.l)], max(a.v, b.v))


def leq(a, b):
    return all(x <= y for x, y in zip(a.l, b.l)) and a.v <= b.v


def const(limbs):
    return B(limbs)


def canonical():
    """any canonical residue (< q, limbs < 2^29) — table entries, decoded inputs"""
    return B([M29] * NL, Q - 1)


# ---- fe29.h ---------------------------------------------------------------------------------
def _fips(cols, vprod):
    """fe_mont_fips on column maxima: the accumulator of column k holds the carry, the operand
    products and the digit products m_i * q_j (m_0 <= 2^29, m_i <= 2^29 - 1); it only grows inside
    a column, so its final value is the value to bound."""
    mmax = [M29 + 1] + [M29] * (NL - 1)
    out = []
    acc = cols[0]
    _check(acc < U64, "column 0 overflows 64 bits")
    acc >>= LB
    for k in range(1, 2 * NL - 1):
        acc += cols[k]
        for i in range(NL):
            j = k - i
            if i < k and 1 <= j < NL:
                acc += mmax[i] * Q29[j]
        _check(acc < U64, "column %d overflows 64 bits" % k)
        if k >= NL:
            out.append(min(acc, M29))
        acc >>= LB
    out[0] += 1
    _check(acc < U32, "top limb of a product exceeds 32 bits")
    out.append(acc)
    # value: (a*b + M*q) / 2^261 with M <= 2^261
    v = (vprod + (1 << RBITS) * Q) >> RBITS
    return B(out, v)


def mul(a, b):
    for x in a.l + b.l:
        _check(x < U32, "multiplier limb exceeds 32 bits")
    c = [0] * 17
    for i in range(NL):
        for j in range(NL):
            c[i + j] += a.l[i] * b.l[j]
    return _fips(c, a.v * b.v)


def sqr(a):
    for x in a.l:
        _check(2 * x < U32, "doubled limb of a squaring exceeds 32 bits")
    return mul(a, a)


def add(a, b):
    r = [x + y for x, y in zip(a.l, b.l)]
    for x in r:
        _check(x < U32, "limb-wise add exceeds 32 bits")
    return B(r, a.v + b.v)


def dbl(a):
    return add(a, a)


def carry(a):
    for x in a.l:
        _check(x < U32, "carry-pass input exceeds 32 bits")
    r = [min(a.l[0], M29)]
    for i in range(1, NL - 1):
        r.append(min(a.l[i], M29) + (a.l[i - 1] >> LB))
    r.append(a.l[NL - 1] + (a.l[NL - 2] >> LB))
    for x in r:
        _check(x < U32, "carry-pass output exceeds 32 bits")
    return B(r, a.v)


def sub_raw(a, b, k):
    """a + (bias_k - b) limb-wise, WITHOUT the carry pass"""
    bias = BIAS[k]
    r = []
    for i in range(NL):
        _check(bias[i] >= b.l[i], "bias%s limb %d (%#x) does not dominate the subtrahend (%#x)"
               % (k, i, bias[i], b.l[i]))
        x = a.l[i] + bias[i]
        _check(x < U32, "biased subtraction exceeds 32 bits")
        r.append(x)
    return B(r, a.v + BIAS_MULT[k] * Q)


def sub(a, b, k):
    return carry(sub_raw(a, b, k))


def equal_ok(a, b):
    """fe_equal(a, b): sub8 then canon (value < 16 q, limbs < 2^31 for the ripple)"""
    d = sub(a, b, 8)
    for x in d.l:
        _check(x < (1 << 31), "fe_ripple input limb exceeds 2^31")
    _check(d.v < 16 * Q, "fe_canon input exceeds 16 q")


# ---- jubjub29.h -----------------------------------------------------------------------------
DBL_SQR = True    # jubjub29.h: DSV_DBL_SQR (2uv as (u+v)^2 - (u^2 + v^2)); False = the multiplication


def ext_double(p):
    uu, vv = sqr(p["u"]), sqr(p["v"])
    zz2 = dbl(sqr(p["z"]))
    vpu = add(vv, uu)
    cu = sub(sqr(add(p["u"], p["v"])), vpu, "4w") if DBL_SQR else dbl(mul(p["u"], p["v"]))
    vmu = sub_raw(vv, uu, 2)
    ct = sub(zz2, vmu, "4w")
    return {"u": mul(cu, ct), "v": mul(vpu, vmu), "z": mul(vmu, ct), "t1": cu, "t2": vpu}


def _add_tail(a, b, c, d):
    cu = sub_raw(b, a, 2)
    cv = add(b, a)
    cz = add(d, c)
    ct = sub(d, c, 2)
    return {"u": mul(cu, ct), "v": mul(cv, cz), "z": mul(cz, ct), "t1": cu, "t2": cv}


def ext_add_niels(p, n):
    a = mul(sub_raw(p["v"], p["u"], 2), n["vmu"])
    b = mul(add(p["v"], p["u"]), n["vpu"])
    c = mul(mul(p["t1"], p["t2"]), n["t2d"])
    d = dbl(mul(p["z"], n["z"]))
    return _add_tail(a, b, c, d)


def ext_add_aniels(p, n):
    a = mul(sub_raw(p["v"], p["u"], 2), n["vmu"])
    b = mul(add(p["v"], p["u"]), n["vpu"])
    c = mul(mul(p["t1"], p["t2"]), n["t2d"])
    d = dbl(p["z"])
    return _add_tail(a, b, c, d)


def ext_add_sub_aniels(p, n):
    """jubjub29.h: ext_add_sub_aniels_t — (p + n, p - n) sharing c, d and z (joint table build)"""
    pm, pp = sub_raw(p["v"], p["u"], 2), add(p["v"], p["u"])
    a, b = mul(pm, n["vmu"]), mul(pp, n["vpu"])
    a2, b2 = mul(pm, n["vpu"]), mul(pp, n["vmu"])
    c = mul(mul(p["t1"], p["t2"]), n["t2d"])
    d = dbl(p["z"])
    cu, cv, cz, ct = sub_raw(b, a, 2), add(b, a), add(d, c), sub(d, c, 2)
    z = mul(cz, ct)
    s_ = {"u": mul(cu, ct), "v": mul(cv, cz), "z": z, "t1": cu, "t2": cv}
    cu2, cv2 = sub_raw(b2, a2, 2), add(b2, a2)
    d_ = {"u": mul(cu2, carry(cz)), "v": mul(cv2, ct), "z": z, "t1": cu2, "t2": cv2}
    return s_, d_


def ext_from_niels(n):
    cu = sub(n["vpu"], n["vmu"], 4)
    cv = add(n["vpu"], n["vmu"])
    d = dbl(n["z"])
    return {"u": mul(cu, d), "v": mul(cv, d), "z": sqr(d), "t1": cu, "t2": cv}


def ext_to_niels(p, d2):
    return {"vpu": carry(add(p["v"], p["u"])), "vmu": sub(p["v"], p["u"], 2), "z": p["z"],
            "t2d": mul(mul(p["t1"], p["t2"]), d2)}


def table_entry(p, d2):
    """What load_var_entry can hand to ext_add_niels for an entry built from p: a negated entry
    swaps vpu/vmu and reads fe_neg2(t2d) (store_var_entry / load_var_entry in common.h), so every
    role must tolerate either bound."""
    n = ext_to_niels(p, d2)
    both = join(n["vpu"], n["vmu"])
    return {"vpu": both, "vmu": both, "z": n["z"],
            "t2d": join(n["t2d"], sub(B([0] * NL), n["t2d"], 2))}


def widen(b):
    """round the bounds up to a coarse grid so that the geometric convergence of the value bounds
    (v -> v*v/R + q) ends in finitely many steps"""
    g = Q >> 8
    limbs = list(b.l)
    limbs[NL - 1] = -(-limbs[NL - 1] // 4096) * 4096
    return B(limbs, -(-b.v // g) * g)


def join_pt(p, q):
    return {k: join(p[k], q[k]) for k in p}


def leq_pt(p, q):
    return all(leq(p[k], q[k]) for k in p)


def prove_group_law(max_rounds=40):
    """Fixpoint of: accumulator -> {double, add niels (table of a variable base), add affine niels
    (fixed-base table)}; niels entries -> to_niels of any accumulator the table build produces.
    Returns the invariant bounds; raises OverflowError_ if any step can overflow."""
    d2 = canonical()
    mont_in = mul(canonical(), canonical())          # fe_to_mont of a decoded coordinate
    acc = {"u": mont_in, "v": mont_in, "z": mont_in, "t1": mont_in, "t2": mont_in}
    fixed = {"vpu": canonical(), "vmu": canonical(), "t2d": canonical()}
    niels = table_entry(acc, d2)
    for rnd in range(max_rounds):
        nxt = join_pt(acc, ext_double(acc))
        # ext_double_affine (table build): z = 1, 2 z^2 = the constant 2
        nxt = join_pt(nxt, ext_double(dict(acc, z=canonical())))   # (ONE is a canonical residue)
        nxt = join_pt(nxt, ext_add_niels(acc, niels))
        nxt = join_pt(nxt, ext_add_aniels(acc, fixed))
        # table build: i*P + P with P's own (affine, z = 1) niels form; chain start: O + entry
        nxt = join_pt(nxt, ext_add_aniels(acc, {k: niels[k] for k in ("vpu", "vmu", "t2d")}))
        # joint table build: the sum / difference pair of one shared mixed addition
        for pt in ext_add_sub_aniels(acc, {k: niels[k] for k in ("vpu", "vmu", "t2d")}):
            nxt = join_pt(nxt, pt)
        nxt = join_pt(nxt, ext_from_niels(niels))
        nn = join_pt(niels, table_entry(nxt, d2))
        if rnd >= 2:
            nxt = {k: widen(v) for k, v in nxt.items()}
            nn = {k: widen(v) for k, v in nn.items()}
        if leq_pt(nxt, acc) and leq_pt(nn, niels):
            # final comparisons of the kernels: ext_eq_affine and the identity test
            equal_ok(acc["u"], mul(mont_in, acc["z"]))
            equal_ok(acc["v"], mul(mont_in, acc["z"]))
            equal_ok(acc["u"], B([0] * NL))
            equal_ok(acc["v"], acc["z"])
            # ext_add_aniels_is_identity: b == a and b + a == d - c inside the last mixed addition
            a_ = mul(sub_raw(acc["v"], acc["u"], 2), fixed["vmu"])
            b_ = mul(add(acc["v"], acc["u"]), fixed["vpu"])
            c_ = mul(mul(acc["t1"], acc["t2"]), fixed["t2d"])
            d_ = dbl(acc["z"])
            equal_ok(b_, a_)
            equal_ok(add(b_, a_), sub(d_, c_, 2))
            return {"acc": acc, "niels": niels, "rounds": rnd}
        acc, niels = nxt, nn
    raise OverflowError_("bounds keep growing: no fixpoint after %d rounds" % max_rounds)


# ---- hades29.h ------------------------------------------------------------------------------
def _load_table(name):
    text = open(_HDR).read()
    m = re.search(r"%s\[[^\]]*\]\[9\] = \{(.*?)\n\};" % name, text, flags=re.S)
    rows = re.findall(r"\{([^}]*)\}", m.group(1))
    return [B([int(x.strip().rstrip("u"), 16) for x in r.split(",")]) for r in rows]


def _load_row(name):
    text = open(_HDR).read()
    m = re.search(r"%s\[9\] = \{([^}]*)\}" % name, text)
    return B([int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")])


def sbox(x):
    x2 = sqr(x)
    x4 = sqr(x2)
    return mul(x4, x)


def mfma_row():
    """hades_mfma.h: a row of a constant linear layer on the matrix cores comes back from the Barrett
    step as 8 words < 2^256 (range proof: gen_constants.py mfma_linear, tests/test_mfma_model.py),
    re-cut into limbs by fe_from_words_plain: limbs 0..7 < 2^29, limb 8 < 2^24"""
    return B([M29] * (NL - 1) + [(1 << (256 - LB * (NL - 1))) - 1], (1 << 256) - 1)


def mfma_operand_ok(x, what):
    """mfma_digits(x): x is cut into 32 bytes by fe_to_words_plain — limbs < 2^29 (limb 0 <= 2^29 for
    an S-box output, which is stored one below its value) and value < 2^256"""
    _check(all(v <= M29 + (1 if i == 0 else 0) for i, v in enumerate(x.l)), what + ": limb exceeds 29 bits")
    _check(x.v < (1 << 256), what + ": value exceeds 2^256")


def hades_full_round(s, rc):
    s = [add(s[k], rc[k]) for k in range(5)]
    s = [sbox(x) for x in s]
    for x in s:
        mfma_operand_ok(x, "S-box output")
    return [mfma_row() for _ in range(5)]


def hades_partial_rounds(s):
    """hades29.h: hades_partial_rounds — every a_r after a_0 is a row, every z_r an S-box output"""
    for x in s:
        mfma_operand_ok(x, "state entering the partial rounds")
    k0 = _load_row("DSV_HADES_K0_HOST")
    a = add(s[4], k0)
    # fe_cond_sub(fe_ripple(.), q): ripple needs limbs < 2^31; the result is < max(q, sum - q)
    for x in a.l:
        _check(x < (1 << 31), "fe_ripple input limb exceeds 2^31")
    a0 = B([M29] * (NL - 1) + [a.v >> (LB * (NL - 1))], max(Q, a.v - Q))
    mfma_operand_ok(a0, "a_0")
    mfma_operand_ok(sbox(a0), "z_0")
    z = sbox(mfma_row())          # rounds 1..58: a_r is a row
    mfma_operand_ok(z, "z_r")
    return [mfma_row() for _ in range(5)]


def hades_permute(s):
    """hades29.h: hades_permute with the ACTUAL round constants (exact limbs) and worst-case state"""
    rc = _load_table("DSV_HADES_RC_HOST")
    s = list(s)
    for r in range(4):
        s = hades_full_round(s, rc[5 * r:5 * r + 5])
    s = hades_partial_rounds(s)
    for r in range(4):
        s = hades_full_round(s, rc[5 * (4 + 59 + r):5 * (4 + 59 + r) + 5])
    return s


def prove_hades():
    """k_challenge (k_hash.hip): the 3-input and the 5-input sponge on worst-case inputs (fe_to_mont of
    any canonical word; the generic first round is the worst case of the constant-folded one), then
    the truncation's fe_from_mont (fe_canon needs < 16 q and limbs < 2^31)."""
    m = mul(canonical(), canonical())
    zero, one = B([0] * NL), canonical()
    out3 = hades_permute([zero, m, m, m, one])
    s = hades_permute([zero, m, m, m, m])
    s[1] = add(s[1], m)
    s[2] = add(s[2], one)
    out5 = hades_permute(s)
    for o in (out3[1], out5[1]):
        h = mul(o, B([1] + [0] * (NL - 1)))
        for x in h.l:
            _check(x < (1 << 31), "fe_ripple input limb exceeds 2^31")
        _check(h.v < 16 * Q, "fe_canon input exceeds 16 q")
    return {"hash3": out3[1], "hash5": out5[1]}


# ---- decode29.h / k_decompress (k_misc.hip) ------------------------------------------------------
def canon_ok(a):
    for x in a.l:
        _check(x < (1 << 31), "fe_ripple input limb exceeds 2^31")
    _check(a.v < 16 * Q, "fe_canon input exceeds 16 q")


def prove_decompress():
    """the straight-line field code of JubJub point decompression on worst-case inputs; every
    power / table product in fe_inv_sqrt and fe_invert is a multiplication of "N" values"""
    one = canonical()
    v = mul(canonical(), canonical())
    v2 = sqr(v)
    num = sub(v2, one, 2)
    den = add(mul(v2, canonical()), one)
    z = mul(num, den)
    n = mul(z, z)                                   # any product of mul outputs / table entries
    n = join(n, sqr(n))
    n = join(n, mul(n, canonical()))
    canon_ok(n)                                     # ts_digit: fe_canon of a power
    u = mul(num, n)
    equal_ok(mul(sqr(u), den), num)
    canon_ok(mul(u, B([1] + [0] * (NL - 1))))       # fe_from_mont(u)
    un = sub(B([0] * NL), u, 2)                     # the other root
    canon_ok(mul(un, B([1] + [0] * (NL - 1))))
    return {"u": u, "num": num, "den": den}


# ---- k_normalize_uvz with inv29.h, k_scalars_from_mont (k_misc.hip), r04 ------------------------
def prove_normalize_and_limb_conversion():
    """the straight-line field code around the r04 additions on worst-case inputs:
    k_scalars_from_mont — fe_canon(fe_mul(plain words, 2^5)); k_normalize_uvz — the running product of
    fe_to_mont(plain z) values, fe_invert_euclid of it (fe_from_mont in, fe_to_mont(plain cofactor)
    and its negation out, or a Fermat power), the walk back down (inv * prefix, inv * z) and the
    quotients plain coordinate x Montgomery 1/z, canonicalised"""
    plain = canonical()                              # eight 32-bit words < q re-cut into 29-bit limbs
    two5 = B([32] + [0] * (NL - 1))
    canon_ok(mul(plain, two5))                       # m = M * 2^5 * 2^-261
    r2 = const(_load("DSV_R2"))
    z = mul(plain, r2)                               # fe_to_mont
    acc = join(const(_load("DSV_ONE")), z)
    acc = join(acc, mul(acc, z))                     # any running product
    one_plain = B([1] + [0] * (NL - 1))
    canon_ok(mul(acc, one_plain))                    # fe_from_mont(acc) inside fe_invert_euclid
    m = mul(canonical(), r2)                         # the cofactor back in Montgomery form
    inv = join(m, sub(B([0] * NL), m, 2))            # ... or its negation (fe_neg2)
    power = mul(acc, acc)
    inv = join(inv, join(power, sqr(power)))         # ... or the Fermat fallback: a product of powers
    zinv = mul(inv, acc)                             # inv * prefix
    inv2 = mul(inv, z)                               # inv * z for the next point
    inv2 = join(inv2, mul(inv2, z))
    zinv = join(zinv, mul(inv2, acc))
    q = mul(plain, zinv)                             # plain coordinate x Montgomery 1/z
    canon_ok(q)
    return {"quotient": q, "inverse": inv}


# ---- k_rlc.hip: batch fast accept, r05 --------------------------------------------
def prove_fast_accept():
    """the field code k_rlc.hip adds to the group law: the curve test and the affine-niels form of a decoded
    point (k_rlc_prep), and the four-waves-per-point operations of k_rlc_scale, which carry tt = t1 * t2
    in the running point — checked as a fixpoint of their own against the niels forms the group-law
    proof admits (every running sum of k_rlc.hip starts from the identity and adds ext_to_niels /
    affine_niels forms: the compositions prove_group_law covers)."""
    inv = prove_group_law()
    d2 = canonical()
    one = const(_load("DSV_ONE"))
    u = v = mul(canonical(), const(_load("DSV_R2")))          # load_fq: fe_to_mont of a decoded coordinate
    # on_curve: 2 v^2 == 2 u^2 + 2 + (2d) u^2 v^2
    uu, vv = sqr(u), sqr(v)
    rhs = mul(mul(uu, vv), d2)
    a = carry(dbl(vv))
    b = carry(add(carry(add(dbl(uu), dbl(one))), rhs))
    equal_ok(a, b)
    # affine_niels(u, v, negate): the forms ext_add_aniels reads must lie inside the group-law invariant
    t = mul(mul(u, v), d2)
    an = {"vpu": join(carry(add(v, u)), sub(v, u, 2)), "t2d": join(t, sub(B([0] * NL), t, 2))}
    an["vmu"] = an["vpu"]
    for k in ("vpu", "vmu", "t2d"):
        _check(leq(an[k], inv["niels"][k]), "affine_niels %s exceeds the niels invariant" % k)
    # xp_double / xp_add: (u, v, z, tt) with tt = t1 * t2 from the idle wave of the second round
    niels = inv["niels"]
    ident = {"u": B([0] * NL), "v": one, "z": one, "tt": B([0] * NL)}

    def xp_double(p):
        uu, vv, zz, s = sqr(p["u"]), sqr(p["v"]), sqr(p["z"]), sqr(add(p["u"], p["v"]))
        zz2, vpu = dbl(zz), add(vv, uu)
        cu = sub(s, vpu, "4w")
        vmu = sub_raw(vv, uu, 2)
        ct = sub(zz2, vmu, "4w")
        return {"u": mul(cu, ct), "v": mul(vpu, vmu), "z": mul(vmu, ct), "tt": mul(cu, vpu)}

    def xp_add(p, n):
        a = mul(sub_raw(p["v"], p["u"], 2), n["vmu"])
        b = mul(add(p["v"], p["u"]), n["vpu"])
        c = mul(p["tt"], n["t2d"])
        d = dbl(mul(p["z"], n["z"]))
        cu, cv, cz, ct = sub_raw(b, a, 2), add(b, a), add(d, c), sub(d, c, 2)
        return {"u": mul(cu, ct), "v": mul(cv, cz), "z": mul(cz, ct), "tt": mul(cu, cv)}

    acc = ident
    for rnd in range(40):
        nxt = join_pt(acc, xp_double(acc))
        nxt = join_pt(nxt, xp_add(acc, niels))
        if rnd >= 2:
            nxt = {k: widen(x) for k, x in nxt.items()}
        if leq_pt(nxt, acc):
            # what leaves the kernel: the identity test and the niels form of a weighted sum
            equal_ok(acc["u"], B([0] * NL))
            equal_ok(acc["v"], acc["z"])
            out = {"vpu": carry(add(acc["v"], acc["u"])), "vmu": sub(acc["v"], acc["u"], 2), "z": acc["z"],
                   "t2d": mul(acc["tt"], d2)}
            for k in out:
                _check(leq(out[k], niels[k]), "weighted sum's %s exceeds the niels invariant" % k)
            return {"acc": acc, "rounds": rnd}
        acc = nxt
    raise OverflowError_("xp bounds keep growing")
