"""Test batches in the reference's IN-MEMORY representation — TEST INFRASTRUCTURE.

The Rust types hold every field element as four u64 Montgomery limbs, R = 2^256
(`BlsScalar(pub [u64; 4])`, `JubJubScalar`, the coordinates of `JubJubExtended`:
/root/reference/Cargo.toml:25-26, src/signatures.rs:58-61, src/keys/public.rs:59).  `mont_case`
builds a signed + tampered batch of one scheme, re-represents every point with a random z
(tests/keys.rs:33-59), converts everything to limbs with PYTHON INTEGERS (independent of both the
oracle's and the engine's arithmetic) and plants the encodings the Rust types cannot hold:
z = 0, a coordinate / message with limbs >= q, a u with limbs >= r.

Also the record layouts a language binding holds (numpy structured dtypes with the field order
and sizes of the Rust structs): `Signature { u: JubJubScalar, R: JubJubExtended }` = 32 + 160 B,
`PublicKey(JubJubExtended)` = 160 B, ... — what the *_mont_cols entry points read in place.
"""
import numpy as np

import harness as H
import oracle_lib as O
import pymodel as M

Q, R_ORDER = M.Q, M.R_ORDER
POINTS = {"single": ("R", "PK"), "double": ("R", "Rp", "PK", "PKp"), "vargen": ("R", "PK", "Gen")}


def limbs_int(x, mod):
    return M.le32((x << 256) % mod)


def to_limbs_py(arr, mod):
    """canonical 32-byte elements [n, 32 k] -> Montgomery limbs, Python integers"""
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    flat = a.reshape(-1, 32)
    out = np.zeros_like(flat)
    for i in range(flat.shape[0]):
        out[i] = np.frombuffer(limbs_int(M.from_le(flat[i]), mod), np.uint8)
    return out.reshape(a.shape)


def mont_case(scheme, n, seed, period=7, plant=True):
    """-> (cols, want): cols = limb arrays in entry-point order (u, points..., m), want = the
    ORACLE's verdicts on those limbs (oracle_verify_*_mont), cross-checked against the oracle's
    verdicts on the canonical projective form (oracle_verify_*_ext) wherever both are defined."""
    rng = np.random.default_rng(seed)
    gen = {"single": O.keygen_sign_single, "double": O.keygen_sign_double,
           "vargen": O.keygen_sign_vargen}[scheme]
    d = gen(n, seed, nthreads=8)
    H.tamper(d, period=period)
    names = POINTS[scheme]
    uvz, ext = {}, {}
    for k in names:
        uvz[k], ext[k] = H.projective(d[k], rng)
    # canonical-domain verdicts; the tamper classes noncanon_u / noncanon_m are encodings >= the
    # modulus, which have no limb form: reduce them (the item then verifies or not on its merits)
    u_c, m_c = d["u"].copy(), d["m"].copy()
    for i in range(n):
        ui, mi = M.from_le(u_c[i]), M.from_le(m_c[i])
        if ui >= R_ORDER:
            u_c[i] = np.frombuffer(M.le32(ui % R_ORDER), np.uint8)
        if mi >= Q:
            m_c[i] = np.frombuffer(M.le32(mi % Q), np.uint8)
    ext_fn = getattr(O, "verify_%s_ext" % scheme)
    want_canonical = ext_fn(u_c, *[ext[k] for k in names], m_c)
    cols = [to_limbs_py(u_c, R_ORDER)] + [to_limbs_py(uvz[k], Q) for k in names] + [to_limbs_py(m_c, Q)]
    planted = []
    if plant and n >= 64:
        # on items that would verify otherwise: z = 0 / a coordinate >= q in each point array,
        # a message >= q, a u >= r — limbs the Rust types cannot hold: verdict 0
        good = [i for i in range(n) if want_canonical[i]]
        it = iter(good[3:])
        for j in range(len(names)):
            i0, i1 = next(it), next(it)
            cols[1 + j][i0, 64:96] = 0                                  # z = 0
            cols[1 + j][i1, 32 * (j % 3):32 * (j % 3) + 32] = 0xFF      # limbs >= q
            planted += [i0, i1]
        im, iu = next(it), next(it)
        cols[-1][im] = np.frombuffer(M.le32(Q), np.uint8)               # exactly q
        cols[0][iu] = np.frombuffer(M.le32(R_ORDER + 5), np.uint8)      # r + 5 (< 2^252.. still >= r)
        planted += [im, iu]
    want = getattr(O, "verify_%s_mont" % scheme)(*cols)
    expect = want_canonical.copy()
    expect[planted] = 0
    assert np.array_equal(want, expect), "oracle: limb form and canonical form disagree"
    return cols, want


# ---- the typed objects of a binding: field order and sizes of the Rust structs ---------------
EXT = [("u", "u1", 32), ("v", "u1", 32), ("z", "u1", 32), ("t1", "u1", 32), ("t2", "u1", 32)]
JUBJUB_EXTENDED = np.dtype([(k, t, s) for k, t, s in EXT])                       # 160 B
RECORDS = {
    # /root/reference/src/signatures.rs:58-61, :180-184, :337-340; src/keys/public.rs:59, :189, :331-334
    "single": (np.dtype([("u", "u1", 32), ("R", JUBJUB_EXTENDED)]), np.dtype([("pk", JUBJUB_EXTENDED)])),
    "double": (np.dtype([("u", "u1", 32), ("R", JUBJUB_EXTENDED), ("R_prime", JUBJUB_EXTENDED)]),
               np.dtype([("pk", JUBJUB_EXTENDED), ("pk_prime", JUBJUB_EXTENDED)])),
    "vargen": (np.dtype([("u", "u1", 32), ("R", JUBJUB_EXTENDED)]),
               np.dtype([("pk", JUBJUB_EXTENDED), ("generator", JUBJUB_EXTENDED)])),
}


def as_records(scheme, cols):
    """-> (sigs, pks, msgs, column views in entry-point order): arrays of records laid out like the
    Rust structs, filled from the dense limb columns (t1, t2 get junk: verify must not read them)"""
    n = cols[0].shape[0]
    sig_t, pk_t = RECORDS[scheme]
    sigs, pks = np.zeros(n, sig_t), np.zeros(n, pk_t)
    msgs = np.ascontiguousarray(cols[-1]).copy()
    sigs["u"] = cols[0]

    def put(rec, field, col):
        rec[field]["u"], rec[field]["v"], rec[field]["z"] = col[:, :32], col[:, 32:64], col[:, 64:]
        rec[field]["t1"], rec[field]["t2"] = 0xA5, 0x5A

    def view(rec, field):
        # u || v || z are the first 96 bytes of the 160-byte JubJubExtended
        base = rec.view(np.uint8).reshape(n, rec.dtype.itemsize)
        off = rec.dtype.fields[field][1]
        return base[:, off:off + 96]

    if scheme == "single":
        put(sigs, "R", cols[1]); put(pks, "pk", cols[2])
        views = [sigs["u"], view(sigs, "R"), view(pks, "pk"), msgs]
    elif scheme == "double":
        put(sigs, "R", cols[1]); put(sigs, "R_prime", cols[2]); put(pks, "pk", cols[3]); put(pks, "pk_prime", cols[4])
        views = [sigs["u"], view(sigs, "R"), view(sigs, "R_prime"), view(pks, "pk"), view(pks, "pk_prime"), msgs]
    else:
        put(sigs, "R", cols[1]); put(pks, "pk", cols[2]); put(pks, "generator", cols[3])
        views = [sigs["u"], view(sigs, "R"), view(pks, "pk"), view(pks, "generator"), msgs]
    return sigs, pks, msgs, views
