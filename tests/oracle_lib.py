"""ctypes binding of oracle/libschnorr_oracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product (schnorr_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
# ORACLE_SO: alternative build of the same source (e.g. `make -C oracle asan` + LD_PRELOAD of
# libasan), never a different implementation
_SO = os.environ.get("ORACLE_SO") or os.path.join(ORACLE_DIR, "libschnorr_oracle.so")


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _SO


_lib = None
_P = ctypes.c_void_p


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_banner.restype = ctypes.c_char_p
        for name in (
            "oracle_verify_single", "oracle_verify_double", "oracle_verify_vargen",
            "oracle_verify_single_ext", "oracle_verify_double_ext", "oracle_verify_vargen_ext",
            "oracle_challenge_single", "oracle_challenge_double",
            "oracle_keygen_sign_single", "oracle_keygen_sign_double",
            "oracle_keygen_sign_vargen", "oracle_scalar_mul", "oracle_fixed_base_entry",
            "oracle_decompress", "oracle_verify_single_wire", "oracle_verify_double_wire",
            "oracle_verify_vargen_wire",
            "oracle_verify_single_mont", "oracle_verify_double_mont", "oracle_verify_vargen_mont",
            "oracle_to_mont", "oracle_from_mont",
        ):
            getattr(L, name).restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def verify_single(u, R, PK, m, nthreads=1):
    u, R, PK, m = map(_u8, (u, R, PK, m))
    n = u.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    lib().oracle_verify_single(_p(u), _p(R), _p(PK), _p(m), ctypes.c_size_t(n), _p(ok),
                               ctypes.c_int(nthreads))
    return ok


def verify_double(u, R, Rp, PK, PKp, m, nthreads=1):
    u, R, Rp, PK, PKp, m = map(_u8, (u, R, Rp, PK, PKp, m))
    n = u.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    lib().oracle_verify_double(_p(u), _p(R), _p(Rp), _p(PK), _p(PKp), _p(m),
                               ctypes.c_size_t(n), _p(ok), ctypes.c_int(nthreads))
    return ok


def verify_vargen(u, R, PK, Gen, m, nthreads=1):
    u, R, PK, Gen, m = map(_u8, (u, R, PK, Gen, m))
    n = u.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    lib().oracle_verify_vargen(_p(u), _p(R), _p(PK), _p(Gen), _p(m), ctypes.c_size_t(n),
                               _p(ok), ctypes.c_int(nthreads))
    return ok


def verify_single_ext(u, R_ext, PK_ext, m):
    u, R_ext, PK_ext, m = map(_u8, (u, R_ext, PK_ext, m))
    n = u.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    lib().oracle_verify_single_ext(_p(u), _p(R_ext), _p(PK_ext), _p(m), ctypes.c_size_t(n),
                                   _p(ok))
    return ok


def verify_double_ext(u, R_ext, Rp_ext, PK_ext, PKp_ext, m):
    u, R_ext, Rp_ext, PK_ext, PKp_ext, m = map(_u8, (u, R_ext, Rp_ext, PK_ext, PKp_ext, m))
    n = u.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    lib().oracle_verify_double_ext(_p(u), _p(R_ext), _p(Rp_ext), _p(PK_ext), _p(PKp_ext), _p(m),
                                   ctypes.c_size_t(n), _p(ok))
    return ok


def verify_vargen_ext(u, R_ext, PK_ext, Gen_ext, m):
    u, R_ext, PK_ext, Gen_ext, m = map(_u8, (u, R_ext, PK_ext, Gen_ext, m))
    n = u.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    lib().oracle_verify_vargen_ext(_p(u), _p(R_ext), _p(PK_ext), _p(Gen_ext), _p(m),
                                   ctypes.c_size_t(n), _p(ok))
    return ok


def _mont(fn, *arrs):
    arrs = [_u8(a) for a in arrs]
    n = arrs[0].shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    getattr(lib(), fn)(*([_p(a) for a in arrs] + [ctypes.c_size_t(n), _p(ok)]))
    return ok


def verify_single_mont(u, R_uvz, PK_uvz, m):
    """every element = the four u64 Montgomery limbs (R = 2^256) the Rust types hold"""
    return _mont("oracle_verify_single_mont", u, R_uvz, PK_uvz, m)


def verify_double_mont(u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m):
    return _mont("oracle_verify_double_mont", u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m)


def verify_vargen_mont(u, R_uvz, PK_uvz, Gen_uvz, m):
    return _mont("oracle_verify_vargen_mont", u, R_uvz, PK_uvz, Gen_uvz, m)


def to_mont(x, fr=False):
    """canonical 32-byte elements [..., 32 k] -> in-memory limbs of the same shape
    (BlsScalar::from_bytes / JubJubScalar::from_bytes); every element must be below the modulus"""
    x = _u8(x)
    out = np.zeros_like(x)
    good = lib().oracle_to_mont(ctypes.c_int(1 if fr else 0), _p(x), ctypes.c_size_t(x.size // 32), _p(out))
    assert good, "to_mont: an element is not below the modulus"
    return out


def from_mont(x, fr=False):
    """in-memory limbs -> canonical bytes (`to_bytes()`); returns (bytes, all_below_modulus)"""
    x = _u8(x)
    out = np.zeros_like(x)
    good = lib().oracle_from_mont(ctypes.c_int(1 if fr else 0), _p(x), ctypes.c_size_t(x.size // 32), _p(out))
    return out, bool(good)


def challenge_single(R, m):
    R, m = _u8(R), _u8(m)
    n = m.shape[0]
    c = np.zeros((n, 32), dtype=np.uint8)
    lib().oracle_challenge_single(_p(R), _p(m), ctypes.c_size_t(n), _p(c))
    return c


def challenge_double(R, Rp, m):
    R, Rp, m = _u8(R), _u8(Rp), _u8(m)
    n = m.shape[0]
    c = np.zeros((n, 32), dtype=np.uint8)
    lib().oracle_challenge_double(_p(R), _p(Rp), _p(m), ctypes.c_size_t(n), _p(c))
    return c


def _wide(rng, n):
    return rng.integers(0, 256, size=(n, 64), dtype=np.uint8)


def keygen_sign_single(n, seed, nthreads=1):
    """benches/signature.rs:48-60 shape: sk, message, then (inside sign) r."""
    rng = np.random.default_rng(seed)
    skw, mw, rw = _wide(rng, n), _wide(rng, n), _wide(rng, n)
    out = {k: np.zeros((n, s), dtype=np.uint8)
           for k, s in (("sk", 32), ("m", 32), ("u", 32), ("R", 64), ("PK", 64))}
    lib().oracle_keygen_sign_single(_p(skw), _p(mw), _p(rw), ctypes.c_size_t(n), _p(out["sk"]),
                                    _p(out["m"]), _p(out["u"]), _p(out["R"]), _p(out["PK"]),
                                    ctypes.c_int(nthreads))
    return out


def keygen_sign_double(n, seed, nthreads=1):
    rng = np.random.default_rng(seed)
    skw, mw, rw = _wide(rng, n), _wide(rng, n), _wide(rng, n)
    out = {k: np.zeros((n, s), dtype=np.uint8)
           for k, s in (("sk", 32), ("m", 32), ("u", 32), ("R", 64), ("Rp", 64), ("PK", 64),
                        ("PKp", 64))}
    lib().oracle_keygen_sign_double(_p(skw), _p(mw), _p(rw), ctypes.c_size_t(n), _p(out["sk"]),
                                    _p(out["m"]), _p(out["u"]), _p(out["R"]), _p(out["Rp"]),
                                    _p(out["PK"]), _p(out["PKp"]), ctypes.c_int(nthreads))
    return out


def keygen_sign_vargen(n, seed, nthreads=1):
    rng = np.random.default_rng(seed)
    skw, gw, mw, rw = _wide(rng, n), _wide(rng, n), _wide(rng, n), _wide(rng, n)
    out = {k: np.zeros((n, s), dtype=np.uint8)
           for k, s in (("sk", 32), ("m", 32), ("u", 32), ("R", 64), ("PK", 64), ("Gen", 64))}
    lib().oracle_keygen_sign_vargen(_p(skw), _p(gw), _p(mw), _p(rw), ctypes.c_size_t(n),
                                    _p(out["sk"]), _p(out["m"]), _p(out["u"]), _p(out["R"]),
                                    _p(out["PK"]), _p(out["Gen"]), ctypes.c_int(nthreads))
    return out


def scalar_mul(scalar, P):
    scalar, P = _u8(scalar), _u8(P)
    n = scalar.shape[0]
    out = np.zeros((n, 64), dtype=np.uint8)
    lib().oracle_scalar_mul(_p(scalar), _p(P), ctypes.c_size_t(n), _p(out))
    return out


def fixed_base_entry(which_gen, window_bits, window, digit):
    out = np.zeros(96, dtype=np.uint8)
    lib().oracle_fixed_base_entry(ctypes.c_int(which_gen), ctypes.c_int(window_bits),
                                  ctypes.c_int(window), ctypes.c_uint32(digit), _p(out))
    return out


def decompress(comp):
    comp = _u8(comp)
    n = comp.shape[0]
    out = np.zeros((n, 64), dtype=np.uint8)
    ok = np.zeros(n, dtype=np.uint8)
    lib().oracle_decompress(_p(comp), ctypes.c_size_t(n), _p(out), _p(ok))
    return out, ok


def _wire(fn, sig, pk, m):
    sig, pk, m = _u8(sig), _u8(pk), _u8(m)
    n = m.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    getattr(lib(), fn)(_p(sig), _p(pk), _p(m), ctypes.c_size_t(n), _p(ok))
    return ok


def verify_single_wire(sig64, pk32, m):
    return _wire("oracle_verify_single_wire", sig64, pk32, m)


def verify_double_wire(sig96, pk64, m):
    return _wire("oracle_verify_double_wire", sig96, pk64, m)


def verify_vargen_wire(sig64, pk64, m):
    return _wire("oracle_verify_vargen_wire", sig64, pk64, m)


def compress(uv):
    """JubJubAffine::to_bytes of affine points [n, 64] -> [n, 32]"""
    uv = _u8(uv)
    out = uv[:, 32:].copy()
    out[:, 31] |= (uv[:, 0] & 1) << 7
    return out
