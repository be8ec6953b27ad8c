"""The matrix-core form of the Hades linear layers (schnorr_amd/csrc/hades_mfma.h) against plain
field arithmetic — CPU only: gen_constants.py re-derives every intermediate of the device code in
integers (digit ranges, biases, the constant folded into the start limbs, the bound of the value
handed to the Montgomery reduction) and builds the operand tables the kernel reads."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "schnorr_amd", "csrc"))
import gen_constants as G  # noqa: E402


def _tables():
    rc, m = G.round_constants(), G.mds()
    A = G.arma_partial_rounds(rc, m)
    return rc, m, A


def test_recurrence_and_dense_layer_models_match_field_arithmetic():
    rc, m, A = _tables()
    atab, starts = G.mfma_recurrence(A["ca"] + A["cz"], A["gamma"])   # asserts inside (self-test + ranges)
    assert len(atab) == 10 * 64 * 16 and len(starts) == G.PARTIAL - 5
    mtab, mstarts = G.mfma_mds(m)
    assert len(mtab) == 5 * 5 * 64 * 16 and len(mstarts) == 5
    assert all(0 <= v < G.Q for v in starts + mstarts)


def test_one_recurrence_round_equals_the_dense_partial_round():
    """a_{r+5} from the byte-matrix product == the S-box input of the dense round r+5"""
    rc, m, A = _tables()
    rec = A["ca"] + A["cz"]
    _, starts = G.mfma_recurrence(rec, A["gamma"])
    rnd = random.Random(5)
    x = [rnd.randrange(G.Q) for _ in range(5)]
    a, z = [], []
    P = G.PARTIAL
    k = [rc[(G.FULL // 2) * 5 + r * 5:(G.FULL // 2) * 5 + (r + 1) * 5] for r in range(P)]
    for r in range(12):
        w = [(x[i] + k[r][i]) % G.Q for i in range(5)]
        a.append(w[4])
        w[4] = pow(w[4], 5, G.Q)
        z.append(w[4])
        x = G._matvec(m, w)
    rinv = pow(G.RMONT, -1, G.Q)
    for r in range(5, 12):
        # operands as the kernel stores them: Montgomery integers (any representative below 2^256),
        # z one below its value
        xs = [a[r - 5 + i] * G.RMONT % G.Q + (G.Q if i & 1 else 0) for i in range(5)]
        xs += [(z[r - 5 + i] * G.RMONT % G.Q or G.Q) - 1 for i in range(5)]
        v = G.mfma_step_model(rec, xs, starts[r - 5])
        assert v < (1 << 256) and v * rinv % G.Q == a[r], r


def test_operand_table_layout():
    """A[m][k] = digit m of c * 2^(8k) mod q: row lane%32, k = 16*(lane/32) + byte"""
    c = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % G.Q
    tab = G.mfma_a_table([c])
    assert len(tab) == 64 * 16
    for lane in (0, 1, 31, 32, 63):
        for byte in (0, 7, 15):
            mrow, kk = lane % 32, 16 * (lane // 32) + byte
            dg = G.balanced_digits(c * (1 << (8 * kk)) % G.Q)
            assert tab[lane * 16 + byte] == dg[mrow] & 255


def test_barrett_step_bound_over_the_whole_range():
    """z - ((z >> 240) * MU >> 32) * q lies in [0, 2^256) for z at the ends of [0, 2^272)"""
    rnd = random.Random(9)
    zs = [0, 1, G.Q - 1, G.Q, (1 << 272) - 1, (1 << 240) - 1, 1 << 240, (1 << 271) + 12345]
    zs += [rnd.randrange(1 << 272) for _ in range(2000)]
    zs += [kq * G.Q + d for kq in (1, 2, 1000, (1 << 17) - 1) for d in (-1, 0, 1)]
    for zv in zs:
        if not 0 <= zv < (1 << 272):
            continue
        qhat = ((zv >> 240) * G.MFMA_MU) >> 32
        r = zv - qhat * G.Q
        assert 0 <= r < (1 << 256), hex(zv)
