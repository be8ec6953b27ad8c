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


def reduce_cols(c):
    """columns 0..8 arrive pre-biased by +M29 (see fe29.h)"""
    k = 0
    c = list(c) + [0]
    for i in range(NL):
        c[i] += M29
        s = c[i] + k
        assert s < U64
        m = (~s) & M29
        assert (s - M29 + m) & M29 == 0 and (s - M29 + m) >> LB == s >> LB
        k = s >> LB
        for j in range(1, NL):
            c[i + j] += m * Q29[j]
            assert c[i + j] < U64, "column overflow in reduction"
            stats["max_col"] = max(stats["max_col"], c[i + j])
    r = []
    for i in range(NL - 1):
        s = c[NL + i] + k
        assert s < U64
        r.append(s & M29)
        k = s >> LB
    assert k < U32
    r.append(k)
    return r


def dot(avec, bvec):
    c = [0] * 17
    for a, b in zip(avec, bvec):
        for l in a + b:
            assert 0 <= l < U32
        for i in range(NL):
            for j in range(NL):
                c[i + j] += a[i] * b[j]
    for x in c:
        assert x < U64, "column overflow in dot product"
        stats["max_col"] = max(stats["max_col"], x)
    return reduce_cols(c)


def dot_plus(avec, bvec, start):
    """fe_dot_const_plus: the column sums start from `start` (constant * R^2 limbs + the M29 bias)"""
    c = [0] * 17
    for a, b in zip(avec, bvec):
        for l in a + b:
            assert 0 <= l < U32
        for i in range(NL):
            for j in range(NL):
                c[i + j] += a[i] * b[j]
    for i in range(NL):
        assert M29 <= start[i] < (1 << 30)
        c[i] += start[i] - M29          # reduce_cols adds the bias itself
    for x in c:
        assert x < U64, "column overflow in dot product"
        stats["max_col"] = max(stats["max_col"], x)
    return reduce_cols(c)


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
def _load_table(name):
    text = open(_HDR).read()
    m = re.search(r"%s\[[^\]]*\]\[9\] = \{(.*?)\n\};" % name, text, flags=re.S)
    rows = re.findall(r"\{([^}]*)\}", m.group(1))
    return [[int(x.strip().rstrip("u"), 16) for x in r.split(",")] for r in rows]


_H = {}


def _hc():
    if not _H:
        _H["rc"] = _load_table("DSV_HADES_RC_HOST")
        _H["mds"] = _load_table("DSV_HADES_MDS_HOST")
        _H["pre"] = _load_table("DSV_HADES_PRE_MDS_HOST")
        _H["k0"] = _load_table("DSV_HADES_KAPPA0_HOST")
        _H["blk"] = _load_table("DSV_HADES_BLOCKS_HOST")
        _H["kf"] = _load_table("DSV_HADES_KFINAL_HOST")
        _H["arma"] = _load_table("DSV_HADES_ARMA_HOST")
        text = open(_HDR).read()
        for name in ("REC", "GAMMA", "FINAL"):
            _H["arma_" + name.lower()] = int(re.search(r"#define DSV_HADES_ARMA_%s (\d+)" % name, text).group(1))
    return _H


def sbox(x):
    x2 = sqr(x)
    x4 = sqr(x2)
    return mul(x4, x)


def hades_full_round(s, rc, mat):
    s = [add(s[k], rc[k]) for k in range(5)]
    s = [sbox(x) for x in s]
    return [dot(s, [mat[k * 5 + j] for j in range(5)]) for k in range(5)]


def hades_partial_rounds_arma(s):
    """hades29.h: hades_partial_rounds_arma, limb-exact"""
    h = _hc()
    k = h["arma"]
    A, Z = [None] * 5, [None] * 5
    A[0] = carry(add(s[4], k[0]))
    Z[0] = sbox(A[0])
    pos = 1
    for r in range(1, 5):
        t = list(s)
        for j in range(r):
            t += [A[j], Z[j]]
        nt = 5 + 2 * r
        A[r] = dot_plus(t, k[pos:pos + nt], k[pos + nt])
        Z[r] = sbox(A[r])
        pos += nt + 1
    assert pos == h["arma_rec"]
    rec = k[h["arma_rec"]:h["arma_rec"] + 10]
    g = h["arma_gamma"]
    for r in range(5, 59):
        p = r % 5
        t = [A[(p + i) % 5] for i in range(5)] + [Z[(p + i) % 5] for i in range(5)]
        A[p] = dot_plus(t, rec, k[g])
        Z[p] = sbox(A[p])
        g += 1
    assert g == h["arma_final"]
    t = [A[4], A[0], A[1], A[2], A[3], Z[4], Z[0], Z[1], Z[2], Z[3]]
    f = h["arma_final"]
    return [dot_plus(t, k[f + 11 * j:f + 11 * j + 10], k[f + 11 * j + 10]) for j in range(5)]


def hades_permute(s, arma=True):
    h = _hc()
    s = list(s)
    if arma:
        for r in range(4):
            s = hades_full_round(s, h["rc"][5 * r:5 * r + 5], h["mds"])
        s = hades_partial_rounds_arma(s)
        for r in range(4):
            s = hades_full_round(s, h["rc"][5 * (4 + 59 + r):5 * (4 + 59 + r) + 5], h["mds"])
        return s
    for r in range(4):
        s = hades_full_round(s, h["rc"][5 * r:5 * r + 5], h["pre"] if r == 3 else h["mds"])
    s[4] = add(s[4], h["k0"][4])
    k = h["blk"]
    pos = 0
    done = 0
    while done < 59:
        lb = min(4, 59 - done)
        z = []
        for m in range(lb):
            z.append(sbox(s[4]))
            a = [s[0], s[1], s[2], s[3]] + z
            row = k[pos:pos + 5 + m]
            s[4] = add(dot(a, row), k[pos + 5 + m])
            pos += 6 + m
        for j in range(4):
            s[j] = dot([s[j]] + z, k[pos:pos + lb + 1])
            pos += lb + 1
        done += lb
    assert pos == len(k)
    for j in range(4):
        s[j] = carry(add(s[j], h["kf"][j]))
    for r in range(4):
        s = hades_full_round(s, h["rc"][5 * (4 + 59 + r):5 * (4 + 59 + r) + 5], h["mds"])
    return s
