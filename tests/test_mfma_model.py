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
    atab, starts = G.mfma_recurrence(A["ca"] + A["cz"], A["gamma"])   # asserts inside (self-test)
    assert len(atab) == 10 * 2 * 64 * 16 and len(starts) == G.PARTIAL - 5
    mtab, mstarts = G.mfma_mds(m)
    assert len(mtab) == 5 * 5 * 2 * 64 * 16 and len(mstarts) == 5


def test_one_recurrence_round_equals_the_dense_partial_round():
    """a_{r+5} from the byte-matrix product == the S-box input of the dense round r+5"""
    rc, m, A = _tables()
    rec = A["ca"] + A["cz"]
    ks = [c * G.RMONT % G.Q for c in rec]
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
        # operands as the kernel stores them: Montgomery integers, z one below
        xs = [a[r - 5 + i] * G.RMONT % G.Q for i in range(5)]
        xs += [(z[r - 5 + i] * G.RMONT % G.Q or G.Q) - 1 for i in range(5)]
        out = G.mfma_step_model(ks, xs, starts[r - 5])
        v = sum(l << (G.LIMB_BITS * i) for i, l in enumerate(out))
        assert v * rinv % G.Q == a[r], r


def test_operand_table_layout():
    """A[m][k] = digit (m - k): row 32*mt + lane%32, k = 16*(lane/32) + byte"""
    kj = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % G.Q
    dg = G.balanced_digits(kj)
    tab = G.mfma_a_table([kj])
    assert len(tab) == 2 * 64 * 16
    for mt in range(2):
        for lane in (0, 1, 31, 32, 63):
            for byte in (0, 7, 15):
                mrow, kk = 32 * mt + lane % 32, 16 * (lane // 32) + byte
                want = dg[mrow - kk] & 255 if 0 <= mrow - kk < 32 else 0
                assert tab[(mt * 64 + lane) * 16 + byte] == want
