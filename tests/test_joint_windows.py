"""The joint 2-bit window table of k_verify_fixed_half (schnorr_amd/csrc/common.h: joint_slot,
recode_signed2, top_digit2): the digit-pair -> signed-slot map read from the source reproduces
da*P + db*R for every pair, and the recoding is exact for both signs.  Integer model only (CPU)."""
import os
import random
import re

import pymodel as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "schnorr_amd", "csrc", "common.h")).read()

# (coefficient of P, coefficient of R) held by each slot, as build_joint_table fills them
SLOT = {1: (1, 0), 2: (0, 1), 3: (2, 0), 4: (0, 2), 5: (1, 1), 6: (1, -1), 7: (2, 1), 8: (2, -1),
        9: (1, 2), 10: (-1, 2), 11: (2, 2)}
A = int("55" * 32, 16)   # bias: digit = 2-bit field - 1, in [-1, 2]
BIAS = 1


def _constants():
    slots = int(re.search(r"slots = (0x[0-9A-Fa-f]+)ULL", SRC).group(1), 16)
    neg = int(re.search(r"\(\((0x[0-9A-Fa-f]+)u >> idx\) & 1u\)", SRC).group(1), 16)
    return slots, neg


def joint_slot(ra, rb):
    slots, neg = _constants()
    idx = ra * 4 + rb
    s = (slots >> (4 * idx)) & 15
    return -s if (neg >> idx) & 1 else s


def recode_signed2(s):
    y = A + s
    assert 0 <= y < 1 << 256
    return y


def test_every_digit_pair_reads_the_right_combination():
    for da in range(-1, 3):
        for db in range(-1, 3):
            s = joint_slot(da + BIAS, db + BIAS)
            if (da, db) == (0, 0):
                assert s == 0
                continue
            i, j = SLOT[abs(s)]
            sign = -1 if s < 0 else 1
            assert (sign * i, sign * j) == (da, db), (da, db, s)
    assert set(abs(joint_slot(a, b)) for a in range(4) for b in range(4)) == set(range(12))


def test_slot_layout_in_the_source_comment_matches():
    line = re.search(r"//\s+slot: (.*)\n", SRC).group(1)
    names = {"P": (1, 0), "R": (0, 1), "2P": (2, 0), "2R": (0, 2), "P+R": (1, 1), "P-R": (1, -1),
             "2P+R": (2, 1), "2P-R": (2, -1), "P+2R": (1, 2), "2R-P": (-1, 2), "2P+2R": (2, 2)}
    for part in line.split("|"):
        k, name = part.split()
        assert SLOT[int(k)] == names[name]


def test_signed_2bit_recoding_is_exact_and_finds_its_top_window():
    rnd = random.Random(11)
    cases = [0, 1, 2, 3, (1 << 128) - 1, 1 << 128, (1 << 160) - 1, (1 << 251) - 1, (1 << 254) - 1]
    cases += [rnd.getrandbits(rnd.randrange(1, 252)) for _ in range(3000)]
    for s in cases:
        y = recode_signed2(s)
        digits = [((y >> (2 * k)) & 3) - BIAS for k in range(128)]
        assert sum(d << (2 * k) for k, d in enumerate(digits)) == s
        nz = y ^ A
        top = (nz.bit_length() - 1) >> 1 if nz else 0
        assert all(d == 0 for d in digits[top + 1:])
        own = (max(s.bit_length(), 1) - 1) // 2
        assert top <= own + 1
        if s.bit_length() & 1:      # odd length: top field 1 -> digit 1 or 2, never a carry out
            assert top == own


def test_window_counts_per_wave():
    """the figure bench.py's work model uses: mean over 64-lane waves of the longest chain"""
    import pymodel as M
    rnd = random.Random(3)
    tops = []
    for _ in range(64 * 60):
        a, b, _neg = M.half_scalars(rnd.getrandbits(250))
        nz = (recode_signed2(a) ^ A) | (recode_signed2(b) ^ A)
        tops.append(((nz.bit_length() - 1) >> 1) + 1)
    waves = [max(tops[i:i + 64]) for i in range(0, len(tops), 64)]
    assert 65.3 < sum(waves) / len(waves) < 66.1
    assert 64.1 < sum(tops) / len(tops) < 64.8


def test_joint_chain_evaluates_a_p_plus_b_r():
    """the chain of k_verify_fixed_half on integers: acc = 4*acc + (da*P + db*R) from the top window"""
    rnd = random.Random(12)
    for _ in range(300):
        a, b = rnd.getrandbits(130), rnd.getrandbits(129)
        bneg = rnd.random() < 0.5
        ya, yb = recode_signed2(a), recode_signed2(b)
        nz = (ya ^ A) | (yb ^ A)
        top = (nz.bit_length() - 1) >> 1 if nz else 0
        P, R0 = rnd.getrandbits(200), rnd.getrandbits(200)  # stand-ins for group elements (Z-module)
        R = -R0 if bneg else R0                             # a negative term negates its point
        acc = 0
        for k in range(top, -1, -1):
            s = joint_slot((ya >> (2 * k)) & 3, (yb >> (2 * k)) & 3)
            i, j = SLOT[abs(s)] if s else (0, 0)
            e = i * P + j * R
            acc = 4 * acc + (-e if s < 0 else e)
        assert acc == a * P + (-b if bneg else b) * R0


def test_joint_chain_on_the_curve_equals_the_reference_equation():
    """the device chain on JubJub points (tests/pymodel.py group law): table slots = i*PK + j*R, chain
    = 4*acc + entry from the top window, plus (b*u)*G: identity exactly when u*G + c*PK == R — incl.
    a key with an order-8 component and tampered signatures."""
    import test_halfgcd as TH
    rnd = random.Random(13)
    t8 = TH.order8_point()
    cases = 0
    for trial in range(12):
        sk, m, r = rnd.randrange(1, M.R_ORDER), rnd.randrange(M.Q), rnd.randrange(1, M.R_ORDER)
        PK = M.pmul(M.GEN, sk)
        R = M.pmul(M.GEN, r)
        if trial % 4 == 3:                      # torsion component in the key (cancels in no R)
            PK = M.padd(PK, t8)
        c = M.challenge(R, m)
        u = (r - c * sk) % M.R_ORDER
        if trial % 3 == 2:
            u = (u + 1) % M.R_ORDER            # tampered
        a, b, bneg = M.half_scalars(c)
        Rt = R if bneg else M.pneg(R)          # the chain adds -|b| * R unless b < 0: table over -R
        table = {k: M.padd(M.pmul(PK, i) if i >= 0 else M.pneg(M.pmul(PK, -i)),
                           M.pmul(Rt, j) if j >= 0 else M.pneg(M.pmul(Rt, -j))) for k, (i, j) in SLOT.items()}
        ya, yb = recode_signed2(a), recode_signed2(b)
        nz = (ya ^ A) | (yb ^ A)
        top = (nz.bit_length() - 1) >> 1 if nz else 0
        acc = M.IDENTITY
        for k in range(top, -1, -1):
            acc = M.pmul(acc, 4)
            sl = joint_slot((ya >> (2 * k)) & 3, (yb >> (2 * k)) & 3)
            if sl:
                acc = M.padd(acc, table[sl] if sl > 0 else M.pneg(table[-sl]))
        w = ((-b if bneg else b) * u) % M.R_ORDER
        acc = M.padd(acc, M.pmul(M.GEN, w))
        want = M.padd(M.pmul(M.GEN, u), M.pmul(PK, c)) == R
        assert (acc == M.IDENTITY) == want, trial
        cases += want
    assert 0 < cases < 12
