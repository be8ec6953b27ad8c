"""Bit-exact Python model of schnorr_amd/csrc/fe29.h + jubjub29.h (limb level), with the
64-bit / 32-bit overflow conditions of the device code turned into assertions.  TEST INFRA.

Every function mirrors the device function of the same name; `Fe` is a list of 9 ints.
"""
import re
import os

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
NL, LB = 9, 29
M29 = (1 << LB) - 1
RBITS = NL * LB
RMONT = (1 << RBITS) % Q
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
QX = {1: _load("DSV_Q29_X1"), 2: _load("DSV_Q29_X2"), 4: _load("DSV_Q29_X4"), 8: _load("DSV_Q29_X8")}
R2 = _load("DSV_R2")
ONE = _load("DSV_ONE")
D2 = _load("DSV_D2")

stats = {"max_col": 0, "max_limb_in": 0}


def val(a):
    return sum(l << (LB * i) for i, l in enumerate(a))


def from_int(x):
    assert 0 <= x < (1 << RBITS)
    return [(x >> (LB * i)) & M29 for i in range(NL)]


def to_mont_int(x):
    return from_int(x * RMONT % Q)


def fips(cols):
    """fe_mont_fips (fe29.h): one 64-bit accumulator over the 17 columns; `cols[k]` = sum of the
    operand products of column k.  The accumulator runs one behind the true column sum from
    column 1 on; result limb 0 takes the missing 1 back."""
    assert len(cols) == 2 * NL - 1
    m = [0] * NL
    r = [0] * NL
    acc = cols[0]
    assert acc < U64
    m[0] = ((~acc) & M29) + 1
    assert (acc + m[0]) & M29 == 0 and (acc + m[0]) >> LB == (acc >> LB) + 1
    acc >>= LB
    true_carry = acc + 1
    for k in range(1, 2 * NL - 1):
        acc += cols[k]
        for i in range(NL):
            j = k - i
            if i < k and 1 <= j < NL:
                acc += m[i] * Q29[j]
        assert acc < U64, "column %d overflows 64 bits" % k
        stats["max_col"] = max(stats["max_col"], acc)
        if k <= NL:
            assert acc + 1 == true_carry + cols[k] + sum(m[i] * Q29[k - i] for i in range(NL)
                                                          if i < k and 1 <= k - i < NL)
        if k < NL:
            m[k] = (~acc) & M29
            assert (acc + 1 + m[k]) & M29 == 0
            true_carry = (acc + 1 + m[k]) >> LB
            assert true_carry == (acc >> LB) + 1
        else:
            r[k - NL] = acc & M29
        acc >>= LB
    r[0] += 1
    assert acc < U32
    r[NL - 1] = acc
    return r


def _cols(pairs):
    c = [0] * (2 * NL - 1)
    for a, b in pairs:
        for l in a + b:
            assert 0 <= l < U32
            stats["max_limb_in"] = max(stats["max_limb_in"], l)
        for i in range(NL):
            for j in range(NL):
                c[i + j] += a[i] * b[j]
    return c


def mul(a, b):
    return fips(_cols([(a, b)]))


def sqr(a):
    for l in a:
        assert 2 * l < U32
    return mul(a, a)


def add(a, b):
    r = [x + y for x, y in zip(a, b)]
    assert all(x < U32 for x in r)
    return r


def dbl(a):
    return add(a, a)


def carry(a):
    r = [a[0] & M29]
    for i in range(1, NL - 1):
        r.append((a[i] & M29) + (a[i - 1] >> LB))
    r.append(a[NL - 1] + (a[NL - 2] >> LB))
    assert all(x < U32 for x in r)
    return r


def sub_raw(a, b, k):
    bias = BIAS[k]
    r = []
    for i in range(NL):
        assert bias[i] >= b[i], "bias does not dominate subtrahend limb %d" % i
        x = a[i] + (bias[i] - b[i])
        assert x < U32
        r.append(x)
    return r


def sub(a, b, k):
    return carry(sub_raw(a, b, k))


def ripple(a):
    r, k = [], 0
    for i in range(NL - 1):
        s = a[i] + k
        assert s < U32
        r.append(s & M29)
        k = s >> LB
    r.append(a[NL - 1] + k)
    return r


def cond_sub(a, m):
    d, borrow = [], 0
    for i in range(NL):
        s = a[i] - m[i] - borrow
        borrow = 1 if s < 0 else 0
        if i < NL - 1:
            assert -(1 << 30) < s < (1 << 30)
            d.append(s & M29)
        else:
            d.append(s)
    return a if borrow else d


def canon(a):
    r = ripple(a)
    assert val(r) < 16 * Q
    for k in (8, 4, 2, 1):
        r = cond_sub(r, QX[k])
    assert val(r) < Q and all(0 <= l <= M29 for l in r)
    return r


def from_mont(a):
    one = [1] + [0] * (NL - 1)
    return canon(mul(a, one))


def to_mont(plain):
    return mul(plain, R2)


def equal(a, b):
    return val(canon(sub(a, b, 8))) == 0


# ---- points ------------------------------------------------------------------------------
def ext_identity():
    return {"u": [0] * NL, "v": list(ONE), "z": list(ONE), "t1": [0] * NL, "t2": [0] * NL}


def ext_from_affine(u, v):
    return {"u": u, "v": v, "z": list(ONE), "t1": u, "t2": v}


DBL_SQR = True    # jubjub29.h: DSV_DBL_SQR (2uv as (u+v)^2 - (u^2 + v^2))


def ext_double(p):
    uu, vv = sqr(p["u"]), sqr(p["v"])
    zz2 = dbl(sqr(p["z"]))
    vpu = add(vv, uu)
    cu = sub(sqr(add(p["u"], p["v"])), vpu, "4w") if DBL_SQR else dbl(mul(p["u"], p["v"]))
    vmu = sub_raw(vv, uu, 2)
    ct = sub(zz2, vmu, "4w")
    return {"u": mul(cu, ct), "v": mul(vpu, vmu), "z": mul(vmu, ct), "t1": cu, "t2": vpu}


def ext_double_affine(u, v):
    """jubjub29.h: doubling of an affine point: 2 z^2 is the constant 2"""
    uu, vv = sqr(u), sqr(v)
    zz2 = dbl(list(ONE))
    vpu = add(vv, uu)
    cu = sub(sqr(add(u, v)), vpu, "4w")
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
    """jubjub29.h: ext_add_sub_aniels_t — (p + n, p - n) sharing c, d and z"""
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


def ext_add_aniels_is_identity(p, n):
    """jubjub29.h: [p + n == O] from the addition's a, b, c, d alone"""
    a = mul(sub_raw(p["v"], p["u"], 2), n["vmu"])
    b = mul(add(p["v"], p["u"]), n["vpu"])
    c = mul(mul(p["t1"], p["t2"]), n["t2d"])
    d = dbl(p["z"])
    return equal(b, a) and equal(add(b, a), sub(d, c, 2))


def ext_from_niels(n):
    cu = sub(n["vpu"], n["vmu"], 4)
    cv = add(n["vpu"], n["vmu"])
    d = dbl(n["z"])
    return {"u": mul(cu, d), "v": mul(cv, d), "z": sqr(d), "t1": cu, "t2": cv}


def ext_to_niels(p):
    return {"vpu": carry(add(p["v"], p["u"])), "vmu": sub(p["v"], p["u"], 2), "z": p["z"],
            "t2d": mul(mul(p["t1"], p["t2"]), D2)}


def affine_of(p):
    zi = pow(val(from_mont(p["z"])), -1, Q)
    return (val(from_mont(p["u"])) * zi % Q, val(from_mont(p["v"])) * zi % Q)


# ---- Hades (hades29.h), with the generated device constants ---------------------------------
