"""Shared test helpers: synthetic batches with a non-trivial expected verdict vector.

Input shape follows the reference's tests/benches (tests/schnorr.rs:16-22,
benches/signature.rs:48-60): sk, message, nonce drawn from one RNG stream; every 16th item is
then corrupted, cycling through tamper classes, so that a kernel
returning a constant verdict cannot pass.
"""
import numpy as np

import pymodel as M

Q, R_ORDER = M.Q, M.R_ORDER
TAMPER_CLASSES = ("wrong_pk", "flip_u", "flip_m", "flip_R", "neg_R", "identity_R", "zero_u",
                  "noncanon_u", "noncanon_m", "swap_R_PK")


def _flip_bit(row, bit):
    row[bit >> 3] ^= np.uint8(1 << (bit & 7))


def tamper(batch, kind_single=True, period=16):
    """In-place corruption of items i = 0 mod period; returns the list of (index, class)."""
    n = batch["u"].shape[0]
    done = []
    k = 0
    for i in range(0, n, period):
        cls = TAMPER_CLASSES[k % len(TAMPER_CLASSES)]
        k += 1
        if cls == "wrong_pk":
            j = (i + 1) % n
            batch["PK"][i] = batch["PK"][j]
        elif cls == "flip_u":
            _flip_bit(batch["u"][i], 3)
        elif cls == "flip_m":
            _flip_bit(batch["m"][i], 77)
        elif cls == "flip_R":
            # replace R by 2R: still a curve point
            j = (i + 2) % n
            batch["R"][i] = batch["R"][j]
        elif cls == "neg_R":
            u = M.from_le(batch["R"][i, :32])
            batch["R"][i, :32] = np.frombuffer(M.le32((-u) % Q), dtype=np.uint8)
        elif cls == "identity_R":
            batch["R"][i, :32] = 0
            batch["R"][i, 32:] = np.frombuffer(M.le32(1), dtype=np.uint8)
        elif cls == "zero_u":
            batch["u"][i] = 0
        elif cls == "noncanon_u":
            # u + r: same residue, but not a canonical JubJubScalar encoding
            u = M.from_le(batch["u"][i])
            batch["u"][i] = np.frombuffer(M.le32(u + R_ORDER), dtype=np.uint8)
        elif cls == "noncanon_m":
            m = M.from_le(batch["m"][i])
            if m + Q < (1 << 256):
                batch["m"][i] = np.frombuffer(M.le32(m + Q), dtype=np.uint8)
            else:
                _flip_bit(batch["m"][i], 5)
        elif cls == "swap_R_PK":
            t = batch["R"][i].copy()
            batch["R"][i] = batch["PK"][i]
            batch["PK"][i] = t
        done.append((i, cls))
    return done


def to_int_point(row):
    return (M.from_le(row[:32]), M.from_le(row[32:]))


def projective(points, rng, zero_z=(), noncanon_z=()):
    """Affine points [n, 64] -> (uvz [n, 96], ext [n, 160]): the same points with a random z != 1 each
    (tests/keys.rs:33-59: equal points, all-different coordinates).  ext = (u, v, z, t1, t2) with
    t1*t2 = uv/z, the oracle's JubJubExtended layout.  Items listed in zero_z get z = 0, items in
    noncanon_z get z + q (same residue, not a canonical encoding): both out of contract, verdict 0."""
    n = points.shape[0]
    uvz = np.zeros((n, 96), np.uint8)
    ext = np.zeros((n, 160), np.uint8)
    for i in range(n):
        uu, vv = to_int_point(points[i])
        z = (int(rng.integers(2, 1 << 62)) * 0x1234567 + int(rng.integers(1, 1 << 62)) * (1 << 190)) % Q
        U, V = uu * z % Q, vv * z % Q
        zenc = z
        if i in zero_z:
            zenc = 0
        elif i in noncanon_z and z + Q < (1 << 256):
            zenc = z + Q
        uvz[i] = np.frombuffer(M.le32(U) + M.le32(V) + M.le32(zenc), np.uint8)
        ext[i] = np.frombuffer(M.le32(U) + M.le32(V) + M.le32(zenc) + M.le32(U) + M.le32(vv), np.uint8)
    return uvz, ext
