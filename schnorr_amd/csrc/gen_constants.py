#!/usr/bin/env python3
"""Generate schnorr_amd/csrc/dsv_constants.h for the HIP engine.

Device field representation ("fe29"): an Fq element is 9 limbs of 29 bits held in 9 x u32
(value = sum l[i] * 2^(29 i)), in Montgomery form with R = 2^261.  This script emits, in that
form: the modulus limbs, the subtraction biases (multiples of q in a redundant limb form whose
every limb dominates the subtrahend's), R^2 for to-Montgomery conversion, the curve constant 2d,
both generators, and the Hades round constants / MDS matrix.

Provenance of the numbers (same as SURVEY.md Appendix A):
  * q, r, d, GENERATOR, GENERATOR_NUMS: dusk-bls12_381 / dusk-jubjub constants — verified
    mathematically by tests/test_constants.py (primality, on-curve, subgroup order).
  * Hades round constants / MDS: recipe documented by dusk-hades (SHA-512 chain from
    b"poseidon-for-plonk"; Cauchy matrix 1/(i + j + 5)).  PARITY UNPINNED: the reference tree
    holds no vector that pins them.

This file is product build tooling; it shares no code with oracle/.
"""
import hashlib
import os

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R_ORDER = 0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7
LIMB_BITS = 29
NLIMBS = 9
MASK = (1 << LIMB_BITS) - 1
RBITS = LIMB_BITS * NLIMBS  # 261
RMONT = (1 << RBITS) % Q

D = (-10240 * pow(10241, -1, Q)) % Q
GEN = (0x3FD2814C43AC65A6F1FBF02D0FD6CCE62E3EBB21FD6C54ED4DF7B7FFEC7BEACA, 0x12)
GEN_NUMS = (
    0x5E67B8F316F414F7BD9514C773FD4456931E316A39FE4541921710179DF76377,
    0x43D80EB3B2F3EB1B7B162DBEEB3B34FD9949BA0F82A5507A6705B707162E3EF8,
)
WIDTH, FULL, PARTIAL = 5, 8, 59


def limbs(x):
    assert 0 <= x < (1 << RBITS)
    return [(x >> (LIMB_BITS * i)) & MASK for i in range(NLIMBS)]


def mont(x):
    return limbs((x % Q) * RMONT % Q)


def bias(k, lift_bits):
    """k*q with limbs 0..7 lifted by 2^lift_bits (borrowed from the limb above)."""
    c = limbs(k * Q)
    b = list(c)
    borrow = 1 << (lift_bits - LIMB_BITS)
    b[0] += 1 << lift_bits
    for i in range(1, NLIMBS - 1):
        b[i] += (1 << lift_bits) - borrow
    b[NLIMBS - 1] -= borrow
    assert sum(v << (LIMB_BITS * i) for i, v in enumerate(b)) == k * Q
    assert all(v >= (1 << lift_bits) - borrow for v in b[:-1]) and b[-1] > 0
    assert all(v < (1 << 32) for v in b)
    return b


def round_constants():
    out, p, data = [], 1, b"poseidon-for-plonk"
    for _ in range((FULL + PARTIAL) * WIDTH):
        data = hashlib.sha512(data).digest()
        p = (int.from_bytes(data, "little") + p) % Q
        out.append(p)
    return out


def mds():
    return [[pow(i + j + WIDTH, -1, Q) for j in range(WIDTH)] for i in range(WIDTH)]


def _matmul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(len(B))) % Q for j in range(len(B[0]))]
            for i in range(len(A))]


def _inv(A):
    n = len(A)
    A = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(A)]
    for c in range(n):
        p = next(r for r in range(c, n) if A[r][c] % Q)
        A[c], A[p] = A[p], A[c]
        iv = pow(A[c][c], -1, Q)
        A[c] = [x * iv % Q for x in A[c]]
        for r in range(n):
            if r != c and A[r][c]:
                f = A[r][c]
                A[r] = [(x - f * y) % Q for x, y in zip(A[r], A[c])]
    return [row[n:] for row in A]


def _matvec(A, v):
    return [sum(A[i][j] * v[j] for j in range(len(v))) % Q for i in range(len(A))]


def arma_partial_rounds(rc, m):
    """The 59 partial rounds as ONE scalar recurrence (hades29.h: hades_partial_rounds_arma).

    Partial round r (dusk-hades order): w = x_r + k_r; a_r = w[4]; z_r = a_r^5; w[4] <- z_r;
    x_{r+1} = M w.  With e = e_4 and g_r = z_r - a_r:  x_{r+1} = M (x_r + k_r) + g_r M e, so the
    S-box inputs are the output of a 5-dimensional linear system driven by g.  By Cayley-Hamilton
    (M^5 = sum c_i M^i) they satisfy
        a_{r+5} = sum_{i<5} c_i a_{r+i} + sum_{t<5} beta_t g_{r+t} + gamma_r,
        beta_t = h_{5-t} - sum_{i>t} c_i h_{i-t},   h_j = e^T M^j e,
    i.e. ten products and ONE reduction per round (a, z = a^5 are the only state), against 11.5
    products and two reductions in the blocked sparse form.  The first five inputs come straight
    from x_0, and x_59 is rebuilt from the last five (a, z) pairs through the observability matrix.
    Returns a dict of plain integers mod q; `selftest` has compared it with the dense rounds."""
    W, P = WIDTH, PARTIAL
    e = [0, 0, 0, 0, 1]
    k = [rc[(FULL // 2) * W + r * W:(FULL // 2) * W + (r + 1) * W] for r in range(P)]
    ident = [[1 if i == j else 0 for j in range(W)] for i in range(W)]
    Mp = [ident]
    for _ in range(6):
        Mp.append(_matmul(Mp[-1], m))
    # characteristic polynomial from the Krylov sequence of e
    K = [[_matvec(Mp[j], e)[i] for j in range(W)] for i in range(W)]      # columns M^j e
    c = _matvec(_inv(K), _matvec(Mp[5], e))
    comb = [[sum(c[i] * Mp[i][a][b] for i in range(W)) % Q for b in range(W)] for a in range(W)]
    assert comb == Mp[5], "Cayley-Hamilton check failed (e_4 not cyclic?)"
    h = [Mp[j][4][4] for j in range(7)]
    beta = [(h[5 - t] - sum(c[i] * h[i - t] for i in range(t + 1, W))) % Q for t in range(W)]
    # kappa_r = e^T (k_r + sum_{s<r} M^{r-s} k_s)
    kappa, acc = [], [0] * W
    for r in range(P):
        wv = [(acc[i] + k[r][i]) % Q for i in range(W)]
        kappa.append(wv[4])
        acc = _matvec(m, wv)
    gamma = [(kappa[r + 5] - sum(c[i] * kappa[r + i] for i in range(W))) % Q for r in range(P - 5)]
    # a_r for r = 1..4:  (e^T M^r) x_0 + sum_{s<r} h_{r-s} (z_s - a_s) + kappa_r
    init = []
    for r in range(1, 5):
        row_x = Mp[r][4][:]
        pairs = [((-h[r - s_]) % Q, h[r - s_]) for s_ in range(r)]            # (coef a_s, coef z_s)
        init.append({"x": row_x, "az": pairs, "const": kappa[r]})
    # x_59 from (a, z)_{54..58}
    b0 = P - 5
    O = [Mp[j][4][:] for j in range(W)]                                        # rows e^T M^j
    Oi = _inv(O)
    H = [[h[j - t] if t < j else 0 for t in range(W)] for j in range(W)]
    kp = []
    acc = [0] * W
    for j in range(W):
        wv = [(acc[i] + k[b0 + j][i]) % Q for i in range(W)]
        kp.append(wv[4])
        acc = _matvec(m, wv)
    M5Oi = _matmul(Mp[5], Oi)
    cols = [[_matvec(Mp[5 - t], e)[i] for t in range(W)] for i in range(W)]    # column t = M^{5-t} e
    M5OiH = _matmul(M5Oi, H)
    G = [[(cols[i][t] - M5OiH[i][t]) % Q for t in range(W)] for i in range(W)]
    Fa = [[(M5Oi[i][t] - G[i][t]) % Q for t in range(W)] for i in range(W)]
    fconst = [(-x) % Q for x in _matvec(M5Oi, kp)]
    for t in range(W):
        v = _matvec(Mp[5 - t], k[b0 + t])
        fconst = [(fconst[i] + v[i]) % Q for i in range(W)]
    out = {"k0": k[0][4], "init": init, "ca": [(c[i] - beta[i]) % Q for i in range(W)], "cz": beta,
           "gamma": gamma, "Fa": Fa, "Fz": G, "fconst": fconst}
    # ---- self-test against the dense definition
    import random as _random
    rnd = _random.Random(7)
    for _ in range(3):
        x0 = [rnd.randrange(Q) for _ in range(W)]
        x, a_ref = list(x0), []
        for r in range(P):
            wv = [(x[i] + k[r][i]) % Q for i in range(W)]
            a_ref.append(wv[4])
            wv[4] = pow(wv[4], 5, Q)
            x = _matvec(m, wv)
        a = [(x0[4] + out["k0"]) % Q]
        z = [pow(a[0], 5, Q)]
        for r in range(1, 5):
            it = init[r - 1]
            v = sum(it["x"][i] * x0[i] for i in range(W)) + it["const"]
            v += sum(ca * a[s_] + cz * z[s_] for s_, (ca, cz) in enumerate(it["az"]))
            a.append(v % Q)
            z.append(pow(a[-1], 5, Q))
        for r in range(5, P):
            v = gamma[r - 5] + sum(out["ca"][i] * a[r - 5 + i] + out["cz"][i] * z[r - 5 + i] for i in range(W))
            a.append(v % Q)
            z.append(pow(a[-1], 5, Q))
        assert a == a_ref, "ARMA recurrence disagrees with the dense partial rounds"
        xf = [(sum(Fa[i][t] * a[b0 + t] + G[i][t] * z[b0 + t] for t in range(W)) + fconst[i]) % Q
              for i in range(W)]
        assert xf == x, "state reconstruction disagrees with the dense partial rounds"
    return out


# ---- the linear layers of Hades on the matrix cores (hades_mfma.h) --------------------------------
# A row  y = sum_j c_j x_j + const  multiplies per-hash values x_j by CONSTANT field elements c_j:
# over a wave that is a constant matrix times a matrix of per-hash columns, which
# v_mfma_i32_32x32x32_i8 evaluates exactly.  Byte k of x_j is multiplied by the constant
# K_jk = c_j * 2^(8k) mod q — the position of the byte is folded into the constant modulo q, so
# every product lands in the SAME 32 byte rows (no Toeplitz band, no empty half tiles):
#     sum_j c_j x_j  ==  sum_m 2^(8m) C_m  (mod q),   C_m = sum_j sum_k digit_m(K_jk) * byte_jk,
# one 32 x 32 tile per term, a 271-bit result, and one Barrett step brings it under 2^256.
# This block builds the operand tables and the start words, proves the ranges for ALL operand
# values from the actual digits, and re-derives every intermediate of the device code in integers.
MFMA_TERMS = 10
MFMA_WBIAS = (1 << 31) + (1 << 47)                                # bias of one 4-row group (two halves)
MFMA_BNET = sum(MFMA_WBIAS << (32 * i) for i in range(8))
MFMA_MU = (1 << 272) // Q                                         # Barrett multiplier for z >> 240


def balanced_digits(x):
    """32 digits in [-128, 127], sum d_k 2^(8k) = x  (x < 2^255: the top digit needs no carry)"""
    out, carry = [], 0
    for k in range(32):
        d = ((x >> (8 * k)) & 255) + carry
        carry = 0
        if d >= 128:
            d -= 256
            carry = 1
        out.append(d)
    assert carry == 0 and sum(d << (8 * k) for k, d in enumerate(out)) == x
    return out


def mfma_consts(coefs):
    """K[j][k] = c_j * 2^(8k) mod q as balanced digits"""
    return [[balanced_digits(c_ * (1 << (8 * k)) % Q) for k in range(32)] for c_ in coefs]


def mfma_a_table(coefs):
    """A operands: [term][lane][16 bytes]; row m = lane%32, k = 16*(lane/32) + byte,
    A[m][k] = digit m of K_jk (operand layout: tools/microbench/mfma_layout.hip)"""
    tab = []
    for Kj in mfma_consts(coefs):
        for lane in range(64):
            m = lane % 32
            for byte in range(16):
                tab.append(Kj[16 * (lane // 32) + byte][m] & 255)
    return tab


def mfma_step_model(coefs, xs, start):
    """mfma_row in integers: xs = the operands as stored (32-byte integers), start = the row's start
    value.  Returns the 256-bit result (8 words on the device)."""
    K = mfma_consts(coefs)
    xd = [[((x >> (8 * k)) & 255) - 128 for k in range(32)] for x in xs]
    z = 0
    for idx in range(8):
        cp = [sum(K[j][k][4 * idx + jj] * xd[j][k] for j in range(len(xs)) for k in range(32)) for jj in range(4)]
        t0 = cp[0] + (cp[1] << 8) + (1 << 31)                       # sign bit flipped: signed -> biased
        t1 = cp[2] + (cp[3] << 8) + (1 << 31)
        assert 0 <= t0 < (1 << 32) and 0 <= t1 < (1 << 32)
        z += (t0 + (t1 << 16)) << (32 * idx)
    z += start
    assert 0 < z < (1 << 272)
    qhat = ((z >> 240) * MFMA_MU) >> 32
    r = z - qhat * Q
    assert 0 <= r < (1 << 256)
    return r


def mfma_linear(coefs, inc, consts):
    """One output row sum_j coefs[j] * x_j + const on the matrix cores, for every const in `consts`
    (plain field elements; the x_j and the result are Montgomery integers).  inc[j] = 1 where the
    stored operand is one below its value.  Returns (A operand bytes, [start value per const])."""
    assert len(coefs) <= 13
    K = mfma_consts(coefs)
    ksum = sum(sum(d << (8 * m) for m, d in enumerate(K[j][k])) for j in range(len(coefs)) for k in range(32))
    corr = (sum(c_ * i_ for c_, i_ in zip(coefs, inc)) + 128 * ksum - MFMA_BNET) % Q
    starts = [(g * RMONT + corr) % Q for g in consts]
    # ---- ranges for ALL operand values (digits in [-128, 127]) from the actual constant digits
    cmax = [sum(abs(K[j][k][m]) for j in range(len(coefs)) for k in range(32)) * 128 for m in range(32)]
    assert max(cmax) < (1 << 23)                                    # int32 accumulators, with room
    for idx in range(8):                                            # halves are signed 32-bit values
        assert cmax[4 * idx] + 256 * cmax[4 * idx + 1] < (1 << 31)
        assert cmax[4 * idx + 2] + 256 * cmax[4 * idx + 3] < (1 << 31)
    spread = sum(c_ << (8 * m) for m, c_ in enumerate(cmax))
    assert MFMA_BNET - spread > 0 and MFMA_BNET + spread + Q < (1 << 272)
    # Barrett step, any z < 2^272: t = z >> 240 < 2^32, qhat = (t * MU) >> 32 with MU = floor(2^272 / q):
    # z/q - 2 - 2^-14.8 < qhat <= z/q, so 0 <= z - qhat * q < (2 + 2^-14) q < 2^256
    assert MFMA_MU < (1 << 18) and 2 * Q + (Q >> 14) < (1 << 256)
    import random as _random
    rnd = _random.Random(11)
    n = len(coefs)
    cases = [[rnd.randrange(1 << 256) for _ in range(n)] for _ in range(4)]
    cases += [[0] * n, [(1 << 256) - 1] * n]
    for sgn in (127, -128):                                          # push the byte rows to their extremes
        xs = []
        for j in range(n):
            # the sign of the top row's digit decides the byte: all rows cannot be extreme at once
            xs.append(sum((((sgn if K[j][k][31] >= 0 else -1 - sgn) + 128) & 255) << (8 * k) for k in range(32)))
        cases.append(xs)
    for xs in cases:
        for gi in (0, len(consts) - 1):
            r = mfma_step_model(coefs, xs, starts[gi])
            want = (sum(c_ * (x + i_) for c_, x, i_ in zip(coefs, xs, inc)) + consts[gi] * RMONT) % Q
            assert r % Q == want, "matrix-core linear-layer model disagrees with the field arithmetic"
    return mfma_a_table(coefs), starts


def mfma_recurrence(rec, gamma):
    """A table + per-round start values for the recurrence; rec = ca + cz (plain field elements);
    the five z operands are stored one below their value (fe_mul's limb 0 is in [1, 2^29])"""
    return mfma_linear(rec, [0] * 5 + [1] * 5, gamma)


def mfma_mds(m):
    """The dense 5 x 5 layer of a full round, one output row at a time: operands are the five
    S-box outputs (stored one below their value), no constant."""
    tab, starts = [], []
    for row in m:
        t_, s_ = mfma_linear(list(row), [1] * WIDTH, [0])
        tab += t_
        starts.append(s_[0])
    return tab, starts


def arr(v):
    return "{" + ", ".join("0x%08xu" % x for x in v) + "}"


def words32(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "dsv_constants.h")
    rc = round_constants()
    m = mds()
    with open(path, "w") as f:
        w = f.write
        w("// GENERATED by schnorr_amd/csrc/gen_constants.py — do not edit.\n")
        w("// fe29 form: 9 x 29-bit limbs in u32, Montgomery R = 2^261 unless noted.\n")
        w("// Hades constants: PARITY UNPINNED (recipe-derived, see generator docstring).\n")
        w("#pragma once\n#include <stdint.h>\n")
        w("#define DSV_HADES_WIDTH %d\n#define DSV_HADES_FULL %d\n#define DSV_HADES_PARTIAL %d\n"
          % (WIDTH, FULL, PARTIAL))
        w("// plain (non-Montgomery) limbs of q\n")
        w("#define DSV_Q29 %s\n" % arr(limbs(Q)))
        w("// q and r as 8 x u32 little-endian words (canonicity checks)\n")
        w("#define DSV_Q32 %s\n" % arr(words32(Q)))
        w("#define DSV_R32 %s\n" % arr(words32(R_ORDER)))
        w("// subtraction biases: k*q, limbs 0..7 lifted to >= 2^30-2\n")
        w("#define DSV_BIAS2 %s\n" % arr(bias(2, 30)))
        w("#define DSV_BIAS4 %s\n" % arr(bias(4, 30)))
        w("#define DSV_BIAS8 %s\n" % arr(bias(8, 30)))
        w("// 4q with limbs 0..7 lifted to >= 2^31-4: dominates an un-carried fe_sub2_raw result\n")
        w("#define DSV_BIAS4W %s\n" % arr(bias(4, 31)))
        w("// canonical multiples of q (fully normalized limbs) for the final reduction\n")
        for k in (1, 2, 4, 8):
            w("#define DSV_Q29_X%d %s\n" % (k, arr(limbs(k * Q))))
        w("// R^2 mod q, plain limbs: to_mont(x) = mont_mul(x, R2)\n")
        w("#define DSV_R2 %s\n" % arr(limbs(RMONT * RMONT % Q)))
        w("#define DSV_ONE %s\n" % arr(mont(1)))
        w("#define DSV_ONE_LIST %s\n" % arr(mont(1))[1:-1])
        w("#define DSV_D2 %s\n" % arr(mont(2 * D)))
        w("#define DSV_D %s\n" % arr(mont(D)))
        w("#define DSV_GEN_U %s\n#define DSV_GEN_V %s\n" % (arr(mont(GEN[0])), arr(mont(GEN[1]))))
        w("#define DSV_GENN_U %s\n#define DSV_GENN_V %s\n"
          % (arr(mont(GEN_NUMS[0])), arr(mont(GEN_NUMS[1]))))
        # square roots (point decompression): q - 1 = 2^32 * t, c = 7^t is a primitive 2^32-th root
        t_odd = (Q - 1) >> 32
        assert pow(7, (Q - 1) // 2, Q) == Q - 1
        e = (t_odd - 1) // 2
        w("#define DSV_SQRT_E_WORDS %s\n" % ("{" + ", ".join("0x%08xu" % ((e >> (32 * i)) & 0xFFFFFFFF) for i in range(7)) + "}"))
        w("#define DSV_SQRT_E_BITS %d\n" % e.bit_length())
        w("#define DSV_ROOT_OF_UNITY %s\n" % arr(mont(pow(7, t_odd, Q))))
        root = pow(7, t_odd, Q)
        assert pow(root, 1 << 31, Q) == Q - 1 and pow(root, 1 << 32, Q) == 1
        # Tonelli-Shanks by 8-bit windows (decode29.h): the discrete log of b = z^t to base g = root
        # is read off in four 8-bit digits; each digit is found by hashing an element of the
        # order-256 subgroup <h>, h = g^(2^24), into a byte table (perfect hash on the two low limbs
        # of the canonical Montgomery form), and cancelled with A_i[d] = g^(-d 2^(8i-1)).
        h = pow(root, 1 << 24, Q)
        hv = [mont(pow(h, d, Q)) for d in range(256)]
        import random as _random
        rnd = _random.Random(20260101)
        HASH_BITS = 13
        while True:
            k1, k2 = rnd.getrandbits(32) | 1, rnd.getrandbits(32) | 1
            idx = [(((v[0] * k1 + v[1] * k2) & 0xFFFFFFFF) >> (32 - HASH_BITS)) for v in hv]
            if len(set(idx)) == 256:
                break
        table = [0] * (1 << HASH_BITS)
        for d, i in enumerate(idx):
            table[i] = d
        w("#define DSV_TS_HASH_K1 0x%08xu\n#define DSV_TS_HASH_K2 0x%08xu\n#define DSV_TS_HASH_BITS %d\n"
          % (k1, k2, HASH_BITS))
        w("// host-side tables: only for the translation units that upload them (dsv_context.hip, k_hash.hip)\n")
        w("#ifdef DSV_HOST_TABLES\n")
        w("static const uint8_t DSV_TS_HASH_HOST[%d] = {\n" % (1 << HASH_BITS))
        for i in range(0, 1 << HASH_BITS, 32):
            w("  " + ", ".join(str(x) for x in table[i:i + 32]) + ",\n")
        w("};\n")
        ginv = pow(root, -1, Q)
        w("static const uint32_t DSV_TS_CANCEL_HOST[4 * 256][9] = {\n")
        for i in range(4):
            for d in range(256):
                e = (d >> 1) if i == 0 else d << (8 * i - 1)
                w("  %s,\n" % arr(mont(pow(ginv, e, Q))))
        w("};\n")
        w("static const uint32_t DSV_HADES_RC_HOST[%d][9] = {\n" % len(rc))
        for c in rc:
            w("  %s,\n" % arr(mont(c)))
        w("};\n")
        w("static const uint32_t DSV_HADES_K0_HOST[9] = %s;  // first partial round's S-box constant\n"
          % arr(mont(arma_partial_rounds(rc, m)["k0"])))
        w("#endif  // DSV_HOST_TABLES\n")
        # first full round of the sponge's first permutation: word 0 (capacity) starts at 0 and, in
        # the 3-input hash, word 4 is the padding 1 — their S-box outputs are constants
        w("// (0 + rc[0])^5 and (1 + rc[4])^5: S-box outputs of the constant words in round 0, one\n")
        w("// below their Montgomery integer (operand convention of hades_mfma.h)\n")
        m1 = lambda x: limbs((x * RMONT % Q or Q) - 1)
        w("#define DSV_HADES_SBOX0_CAP_M1 %s\n" % arr(m1(pow(rc[0], 5, Q))))
        w("#define DSV_HADES_SBOX0_PAD_M1 %s\n" % arr(m1(pow(1 + rc[4], 5, Q))))
        # ---- partial rounds as one scalar recurrence (arma_partial_rounds)
        A = arma_partial_rounds(rc, m)
        rec = A["ca"] + A["cz"]
        # ---- the same recurrence on the matrix cores (hades_mfma.h)
        atab, starts = mfma_recurrence(rec, A["gamma"])
        words_ = [atab[i] | atab[i + 1] << 8 | atab[i + 2] << 16 | atab[i + 3] << 24 for i in range(0, len(atab), 4)]
        w("// recurrence on the matrix cores (generator: mfma_recurrence): A operands [term][lane][4 words]\n")
        w("#define DSV_HADES_MFMA_A_WORDS %d\n" % len(words_))
        w("#define DSV_HADES_MFMA_A_LIST \\\n")
        for i in range(0, len(words_), 8):
            w("  " + ", ".join("0x%08xu" % x for x in words_[i:i + 8]) + (", \\\n" if i + 8 < len(words_) else "\n"))
        w("#define DSV_HADES_MFMA_ROUNDS %d\n" % len(starts))
        w("#define DSV_HADES_MFMA_START_LIST \\\n")
        for i, st in enumerate(starts):
            w("  " + arr(words32(st)) + (", \\\n" if i + 1 < len(starts) else "\n"))
        w("#define DSV_HADES_MFMA_MU 0x%xu\n" % MFMA_MU)
        # ---- the dense layer of the full rounds, same machinery: [row][term][lane][4 words]
        mtab, mstarts = mfma_mds(m)
        words_ = [mtab[i] | mtab[i + 1] << 8 | mtab[i + 2] << 16 | mtab[i + 3] << 24 for i in range(0, len(mtab), 4)]
        w("#define DSV_HADES_MFMA_MDS_WORDS %d\n" % len(words_))
        w("#define DSV_HADES_MFMA_MDS_LIST \\\n")
        for i in range(0, len(words_), 8):
            w("  " + ", ".join("0x%08xu" % x for x in words_[i:i + 8]) + (", \\\n" if i + 8 < len(words_) else "\n"))
        w("#define DSV_HADES_MFMA_MDS_START_LIST \\\n")
        for i, st in enumerate(mstarts):
            w("  " + arr(words32(st)) + (", \\\n" if i + 1 < len(mstarts) else "\n"))
        # ---- start-up rows (a_1..a_4) and state rebuild (5 rows) of the recurrence, same machinery;
        # operand order as in hades_partial_rounds_arma: x_0 (5), then (a_s, z_s) pairs / a (5), z (5)
        etab, estarts = [], []
        for r_, it in enumerate(A["init"], start=1):
            cs = it["x"] + [v for pair in it["az"] for v in pair]
            t_, s_ = mfma_linear(cs, [0] * 5 + [0, 1] * r_, [it["const"]])
            etab += t_
            estarts.append(s_[0])
        for j in range(WIDTH):
            t_, s_ = mfma_linear(A["Fa"][j] + A["Fz"][j], [0] * 5 + [1] * 5, [A["fconst"][j]])
            etab += t_
            estarts.append(s_[0])
        words_ = [etab[i] | etab[i + 1] << 8 | etab[i + 2] << 16 | etab[i + 3] << 24 for i in range(0, len(etab), 4)]
        w("// start-up rows (7, 9, 11, 13 terms) then the five 10-term rows of the state rebuild\n")
        w("#define DSV_HADES_MFMA_EDGE_WORDS %d\n" % len(words_))
        w("#define DSV_HADES_MFMA_EDGE_LIST \\\n")
        for i in range(0, len(words_), 8):
            w("  " + ", ".join("0x%08xu" % x for x in words_[i:i + 8]) + (", \\\n" if i + 8 < len(words_) else "\n"))
        w("#define DSV_HADES_MFMA_EDGE_START_LIST \\\n")
        for i, st in enumerate(estarts):
            w("  " + arr(words32(st)) + (", \\\n" if i + 1 < len(estarts) else "\n"))
    print("wrote", path)


if __name__ == "__main__":
    main()
