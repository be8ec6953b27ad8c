"""Pure-Python (big-int) model of the dusk-schnorr verify/sign path.  TEST INFRASTRUCTURE.

Independent second restatement used to cross-check the C oracle (oracle/schnorr_oracle.c) on
small cases: it uses the *affine* complete twisted-Edwards addition law and Python's modular
inverse, not the extended-coordinate formulas of the oracle, so a slip in either shows up as a
disagreement.  Follows the same reference call sites:

  verify            /root/reference/src/keys/public.rs:121-130, 222-244, 401-415
  challenge hashes  /root/reference/src/signatures.rs:127-134, 275-290
  sign              /root/reference/src/keys/secret.rs:150-168, 217-240, 433-451

Hash constants: same published recipe as oracle/gen_constants.py (PARITY UNPINNED).
"""
import hashlib

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R_ORDER = 0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7
D = (-10240 * pow(10241, -1, Q)) % Q
GEN = (0x3FD2814C43AC65A6F1FBF02D0FD6CCE62E3EBB21FD6C54ED4DF7B7FFEC7BEACA, 0x12)
GEN_NUMS = (
    0x5E67B8F316F414F7BD9514C773FD4456931E316A39FE4541921710179DF76377,
    0x43D80EB3B2F3EB1B7B162DBEEB3B34FD9949BA0F82A5507A6705B707162E3EF8,
)
IDENTITY = (0, 1)
WIDTH, FULL, PARTIAL = 5, 8, 59


def on_curve(p):
    u, v = p
    return (-u * u + v * v - 1 - D * u * u * v * v) % Q == 0


def padd(p, q):
    u1, v1 = p
    u2, v2 = q
    k = D * u1 * u2 * v1 * v2 % Q
    u3 = (u1 * v2 + v1 * u2) * pow(1 + k, -1, Q) % Q
    v3 = (v1 * v2 + u1 * u2) * pow(1 - k, -1, Q) % Q
    return (u3, v3)


def pneg(p):
    return ((-p[0]) % Q, p[1])


def pmul(p, k):
    acc = IDENTITY
    for bit in range(k.bit_length() - 1, -1, -1):
        acc = padd(acc, acc)
        if (k >> bit) & 1:
            acc = padd(acc, p)
    return acc


_RC = None
_MDS = None


def _consts():
    global _RC, _MDS
    if _RC is None:
        rc, p, data = [], 1, b"poseidon-for-plonk"
        for _ in range((FULL + PARTIAL) * WIDTH):
            data = hashlib.sha512(data).digest()
            p = (int.from_bytes(data, "little") + p) % Q
            rc.append(p)
        _RC = rc
        _MDS = [[pow(i + j + WIDTH, -1, Q) for j in range(WIDTH)] for i in range(WIDTH)]
    return _RC, _MDS


def hades_permute(state):
    rc, mds = _consts()
    s = list(state)
    ci = 0
    for rnd in range(FULL + PARTIAL):
        full = rnd < FULL // 2 or rnd >= FULL // 2 + PARTIAL
        s = [(x + rc[ci + k]) % Q for k, x in enumerate(s)]
        ci += WIDTH
        if full:
            s = [pow(x, 5, Q) for x in s]
        else:
            s[4] = pow(s[4], 5, Q)
        s = [sum(mds[k][j] * s[j] for j in range(WIDTH)) % Q for k in range(WIDTH)]
    return s


def sponge_hash(msgs):
    st = [0] * WIDTH
    rate = WIDTH - 1
    chunks = [msgs[i : i + rate] for i in range(0, len(msgs), rate)]
    for ci, ch in enumerate(chunks):
        for k, x in enumerate(ch):
            st[1 + k] = (st[1 + k] + x) % Q
        if ci == len(chunks) - 1:
            if len(ch) < rate:
                st[len(ch) + 1] = (st[len(ch) + 1] + 1) % Q
            else:
                st = hades_permute(st)
                st[1] = (st[1] + 1) % Q
        st = hades_permute(st)
    return st[1]


def truncated_hash(msgs):
    return sponge_hash(msgs) & ((1 << 250) - 1)


def challenge(R, m):
    return truncated_hash([R[0], R[1], m])


def challenge_double(R, Rp, m):
    return truncated_hash([R[0], R[1], Rp[0], Rp[1], m])


def verify_single(u, R, PK, m):
    c = challenge(R, m)
    return padd(pmul(GEN, u), pmul(PK, c)) == R


def verify_double(u, R, Rp, PK, PKp, m):
    c = challenge_double(R, Rp, m)
    return padd(pmul(GEN, u), pmul(PK, c)) == R and padd(pmul(GEN_NUMS, u), pmul(PKp, c)) == Rp


def verify_vargen(u, R, PK, Gen, m):
    c = challenge(R, m)
    return padd(pmul(Gen, u), pmul(PK, c)) == R


def sign_single(sk, m, r):
    R = pmul(GEN, r)
    c = challenge(R, m)
    return (r - c * sk) % R_ORDER, R


def sign_double(sk, m, r):
    R, Rp = pmul(GEN, r), pmul(GEN_NUMS, r)
    c = challenge_double(R, Rp, m)
    return (r - c * sk) % R_ORDER, R, Rp


def sign_vargen(sk, gen, m, r):
    R = pmul(gen, r)
    c = challenge(R, m)
    return (r - c * sk) % R_ORDER, R


def compress(p):
    u, v = p
    return (v | ((u & 1) << 255)).to_bytes(32, "little")


def le32(x):
    return int(x).to_bytes(32, "little")


def from_le(b):
    return int.from_bytes(bytes(b), "little")


def point_bytes(p):
    return le32(p[0]) + le32(p[1])


# ---- half-size scalars (schnorr_amd/csrc/halfgcd.h), integer model -------------------------
def half_scalars(c):
    """(a, b_mag, b_neg) with a = b*c (mod 8r), b odd; same step sequence as the device code."""
    N = 8 * R_ORDER
    A, B, tA, tB, neg = N, c, 0, 1, False
    while B >= (1 << 128):
        if A >= B:
            k = A.bit_length() - B.bit_length()
            if (B << k) > A:
                k -= 1
            A -= B << k
            tA += tB << k
        else:
            A, B, tA, tB = B, A, tB, tA
            neg = not neg
    if tB & 1:
        return B, tB, neg
    # tB even => tA odd, and so is tA + k*tB: pick k where the two components balance
    best = (A, tA)
    k0 = max(0, (A - tA) // (B + tB))
    for kk in (k0, k0 + 1):
        a, b = A - kk * B, tA + kk * tB
        if 1 <= k0 < 2147483000 and a >= 0 and b < (1 << 160) and \
                max(a.bit_length(), b.bit_length()) < max(best[0].bit_length(), best[1].bit_length()):
            best = (a, b)
    return best[0], best[1], not neg


def verify_single_half(u, R, PK, m):
    """(b*u mod r)*G + a*PK - b*R == O  — must equal verify_single on every on-curve input."""
    c = challenge(R, m)
    a, b, bn = half_scalars(c)
    sb = -b if bn else b
    w = (sb * u) % R_ORDER
    t = padd(pmul(GEN, w), pmul(PK, a))
    t = padd(t, pmul(R, b) if bn else pmul(pneg(R), b))
    return t == IDENTITY


# ---- inversion by extended Euclid (schnorr_amd/csrc/inv29.h), the device's step rules -------------
def _image(x):
    """double-precision image of a non-negative integer as to_double8 builds it (Horner over 32-bit
    words, one rounding per step); only its relative accuracy (< 2^-50) matters to the algorithm"""
    d = 0.0
    for i in range(7, -1, -1):
        d = d * 4294967296.0 + float((x >> (32 * i)) & 0xFFFFFFFF)
    return d


def inv_euclid(x, mod=Q, max_iter=1024):
    """(inverse or None, half_steps, fell_back): the algorithm of inv29.h on Python integers and floats —
    quotient estimates floor(dX / dY * (1 - 2^-30)) from the images, alternating roles, exact compare
    when the images are within 2^-28, stop when one side is 0; anything irregular (x = 0, a quotient
    that does not fit 31 bits, the cap) reports a fall-back (the device then runs Fermat).  Raises if an
    estimate ever exceeds the true quotient (the subtraction would borrow) or a cofactor passes 2^256."""
    A, B, tA, tB = mod, x, 0, 1
    steps = 0

    def step(X, tX, Y, tY):
        qd = _image(X) / _image(Y) * (1.0 - 2.0 ** -30)
        if not qd < 2147483647.0:
            return None
        qe = int(qd)
        if qe == 0 and _image(X) >= _image(Y) * (1.0 - 2.0 ** -28) and X >= Y:
            qe = 1
        assert qe * Y <= X, "estimate above the true quotient"
        tX += qe * tY
        assert tX < (1 << 256)
        return X - qe * Y, tX

    it = 0
    ok = True
    while it < max_iter and ok and B != 0:
        r = step(A, tA, B, tB)
        steps += 1
        if r is None:
            ok = False
            break
        A, tA = r
        if A == 0:
            break
        r = step(B, tB, A, tA)
        steps += 1
        if r is None:
            ok = False
            break
        B, tB = r
        it += 1
    on_b = A == 0
    good = ok and it < max_iter and ((on_b and B == 1) or (not on_b and B == 0 and A == 1))
    if not good:
        return (pow(x, -1, mod) if x % mod else 0), steps, True
    t = tB if on_b else tA
    assert t < mod
    return (t if on_b else (mod - t) % mod), steps, False


# ---- three short scalars for the var-generator equation (schnorr_amd/csrc/lattice3.h) --------
def lattice3(u, c, tbound=float(1 << 31), max_batches=24, max_passes=40):
    """(x, y, z): x = z*u, y = z*c (mod 8r), z odd — the device algorithm on Python integers and
    floats: greedy pairwise reduction of (8r,0,0), (0,8r,0), (u,c,1), Lehmer style (passes on
    double-precision images, the accumulated transformation applied exactly per batch)."""
    N = 8 * R_ORDER
    B = [[N, 0, 0], [0, N, 0], [u, c, 1]]
    pairs = ((0, 1), (0, 2), (1, 2), (1, 0), (2, 0), (2, 1))
    for _ in range(max_batches):
        D = [[float(x) for x in v] for v in B]
        T = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
        any_change = stop = False
        for _p in range(max_passes):
            changed = False
            for i, j in pairs:
                njj = D[j][0] * D[j][0] + D[j][1] * D[j][1] + D[j][2] * D[j][2]
                dij = D[i][0] * D[j][0] + D[i][1] * D[j][1] + D[i][2] * D[j][2]
                q = float(round(dij / njj)) if njj > 0 else 0.0
                nt = [a - q * b for a, b in zip(T[i], T[j])]
                ok = max(abs(x) for x in nt) < tbound and not stop
                if q != 0 and not ok:
                    stop = True
                if q != 0 and ok:
                    T[i] = nt
                    D[i] = [a - q * b for a, b in zip(D[i], D[j])]
                    changed = True
            any_change |= changed
            if stop or not changed:
                break
        if not any_change:
            break
        Ti = [[int(x) for x in r] for r in T]
        B = [[sum(Ti[i][k] * B[k][m] for k in range(3)) for m in range(3)] for i in range(3)]
    best, blen = (u, c, 1), 252
    for v in B:
        ln = max(abs(x).bit_length() for x in v)
        if v[2] & 1 and ln < blen:
            best, blen = tuple(v), ln
    return best


def verify_vargen(u, R, PK, Gen, m):
    """/root/reference/src/keys/public.rs:401-415: u*Gen + c*PK == R with c = H(R, m)"""
    c = challenge(R, m)
    return padd(pmul(Gen, u), pmul(PK, c)) == R


def verify_vargen_lattice(u, R, PK, Gen, m):
    """x*Gen + y*PK - z*R == O — must equal verify_vargen on every on-curve input."""
    c = challenge(R, m)
    x, y, z = lattice3(u, c)

    def smul(p, k):
        return pmul(p, k) if k >= 0 else pmul(pneg(p), -k)

    t = padd(padd(smul(Gen, x), smul(PK, y)), smul(R, -z))
    return t == IDENTITY
