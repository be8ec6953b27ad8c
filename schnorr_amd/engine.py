"""Batch entry points of the MI355X engine over numpy (host buffers) and torch (HBM-resident)
arrays.  Thin marshalling only — all arithmetic happens in libdsv.so's HIP kernels.

Array conventions (see include/dsv.h): uint8, C-contiguous,
  scalars  [n, 32]   points [n, 64] (affine u || v)   ext points [n, 96] (u || v || z)
Returns uint8 verdict vectors [n] (1 = the reference's verify() would return true).
"""
import ctypes

import numpy as np

from . import _lib

_initialised = set()


def init(device=0):
    """dsv_init: create the context of one GPU (fixed-base tables, streams).  Idempotent; several
    devices may be initialised in one process."""
    if device in _initialised:
        return
    L = _lib.load()
    _lib.check(L.dsv_init(ctypes.c_int(device)))
    _initialised.add(device)


def init_visible():
    """dsv_init_visible: every device listed in $DSV_DEVICES, else every visible one; returns the
    list of initialised ordinals."""
    rc = _lib.load().dsv_init_visible()
    if rc < 0:
        _lib.check(rc)
    devs = initialized_devices()
    _initialised.update(devs)
    return devs


def set_device(device):
    """Device of this thread's host-buffer entry points (default: the first one initialised)."""
    _lib.check(_lib.load().dsv_set_device(ctypes.c_int(device)))


def initialized_devices():
    buf = (ctypes.c_int * 16)()
    n = _lib.load().dsv_initialized_devices(buf, 16)
    return [buf[i] for i in range(min(n, 16))]


def device_numa(device=0):
    """dsv_device_numa: {"bdf": PCI address, "node": NUMA node of the device's PCIe root (-1: unknown or
    DSV_NUMA=0), "cpus": that node's cpus} — where the device's copy threads are bound"""
    cpus = (ctypes.c_int * 1024)()
    node = ctypes.c_int(-1)
    bdf = ctypes.create_string_buffer(32)
    n = _lib.load().dsv_device_numa(ctypes.c_int(device), ctypes.byref(node), cpus, 1024, bdf)
    if n < 0:
        _lib.check(n)
    return {"bdf": bdf.value.decode(), "node": node.value, "cpus": [cpus[i] for i in range(min(n, 1024))]}


def numa_lookup(sysfs_root, bdf):
    """dsv_debug_numa_lookup: (node, cpus) of a PCI device under a sysfs tree; no device needed"""
    cpus = (ctypes.c_int * 4096)()
    node = ctypes.c_int(-1)
    n = _lib.load().dsv_debug_numa_lookup(sysfs_root.encode(), bdf.encode(), ctypes.byref(node), cpus, 4096)
    if n < 0:
        _lib.check(n)
    return node.value, [cpus[i] for i in range(min(n, 4096))]


def shutdown(device=None):
    """dsv_shutdown (all devices) or dsv_shutdown_device."""
    if not _initialised:
        return
    if device is None:
        _lib.check(_lib.load().dsv_shutdown())
        _initialised.clear()
    else:
        _lib.check(_lib.load().dsv_shutdown_device(ctypes.c_int(device)))
        _initialised.discard(device)


def set_host_threads(n):
    """Copy threads of the host entry points (0 = default); returns the value in force."""
    return int(_lib.load().dsv_set_host_threads(ctypes.c_int(n)))


def version():
    return _lib.load().dsv_version().decode()


def _arr(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 1 and a.shape[0] == width:
        a = a.reshape(1, width)
    if a.ndim != 2 or a.shape[1] != width:
        raise ValueError("expected uint8 array of shape [n, %d], got %r" % (width, a.shape))
    return a


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _same_n(*arrs):
    n = arrs[0].shape[0]
    for a in arrs:
        if a.shape[0] != n:
            raise ValueError("batch arrays disagree on n: %r" % [x.shape for x in arrs])
    return n


# ------------------------------------------------------------------ host-buffer path
def verify_single(u, R, PK, m):
    u, R, PK, m = _arr(u, 32), _arr(R, 64), _arr(PK, 64), _arr(m, 32)
    n = _same_n(u, R, PK, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_single(_p(u), _p(R), _p(PK), _p(m), ctypes.c_size_t(n), _p(ok)))
    return ok


def verify_double(u, R, Rp, PK, PKp, m):
    u, R, Rp, PK, PKp, m = (_arr(u, 32), _arr(R, 64), _arr(Rp, 64), _arr(PK, 64), _arr(PKp, 64),
                            _arr(m, 32))
    n = _same_n(u, R, Rp, PK, PKp, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_double(_p(u), _p(R), _p(Rp), _p(PK), _p(PKp), _p(m),
                                             ctypes.c_size_t(n), _p(ok)))
    return ok


def verify_vargen(u, R, PK, Gen, m):
    u, R, PK, Gen, m = _arr(u, 32), _arr(R, 64), _arr(PK, 64), _arr(Gen, 64), _arr(m, 32)
    n = _same_n(u, R, PK, Gen, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_vargen(_p(u), _p(R), _p(PK), _p(Gen), _p(m),
                                             ctypes.c_size_t(n), _p(ok)))
    return ok


def verify_single_multi(u, R, PK, m):
    """dsv_verify_single_multi: one host batch sharded over every initialised device."""
    u, R, PK, m = _arr(u, 32), _arr(R, 64), _arr(PK, 64), _arr(m, 32)
    n = _same_n(u, R, PK, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_single_multi(_p(u), _p(R), _p(PK), _p(m), ctypes.c_size_t(n),
                                                   _p(ok)))
    return ok


def verify_double_multi(u, R, Rp, PK, PKp, m):
    u, R, Rp, PK, PKp, m = (_arr(u, 32), _arr(R, 64), _arr(Rp, 64), _arr(PK, 64), _arr(PKp, 64),
                            _arr(m, 32))
    n = _same_n(u, R, Rp, PK, PKp, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_double_multi(_p(u), _p(R), _p(Rp), _p(PK), _p(PKp), _p(m),
                                                   ctypes.c_size_t(n), _p(ok)))
    return ok


def verify_vargen_multi(u, R, PK, Gen, m):
    u, R, PK, Gen, m = _arr(u, 32), _arr(R, 64), _arr(PK, 64), _arr(Gen, 64), _arr(m, 32)
    n = _same_n(u, R, PK, Gen, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_vargen_multi(_p(u), _p(R), _p(PK), _p(Gen), _p(m),
                                                   ctypes.c_size_t(n), _p(ok)))
    return ok


def to_hash_inputs(uvz):
    """JubJubExtended::to_hash_inputs over [n, 96] (u || v || z) -> ([n, 64] affine, ok[n])."""
    uvz = _arr(uvz, 96)
    n = uvz.shape[0]
    out = np.zeros((n, 64), dtype=np.uint8)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_to_hash_inputs(_p(uvz), ctypes.c_size_t(n), _p(out), _p(ok)))
    return out, ok


def _ext_call(name, widths, arrays):
    arrs = [_arr(a, w) for a, w in zip(arrays, widths)]
    n = _same_n(*arrs)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(getattr(_lib.load(), name)(*([_p(a) for a in arrs] + [ctypes.c_size_t(n), _p(ok)])))
    return ok


def verify_single_ext(u, R_uvz, PK_uvz, m, multi=False):
    """Projective points (u || v || z, 96 B): the device does to_hash_inputs."""
    return _ext_call("dsv_verify_single_ext" + ("_multi" if multi else ""), (32, 96, 96, 32),
                     (u, R_uvz, PK_uvz, m))


def verify_double_ext(u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m, multi=False):
    return _ext_call("dsv_verify_double_ext" + ("_multi" if multi else ""), (32, 96, 96, 96, 96, 32),
                     (u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m))


def verify_vargen_ext(u, R_uvz, PK_uvz, Gen_uvz, m, multi=False):
    return _ext_call("dsv_verify_vargen_ext" + ("_multi" if multi else ""), (32, 96, 96, 96, 32),
                     (u, R_uvz, PK_uvz, Gen_uvz, m))


# ---- the reference's in-memory representation: every element = four u64 Montgomery limbs (R = 2^256)
def verify_single_mont(u, R_uvz, PK_uvz, m, multi=False):
    return _ext_call("dsv_verify_single_mont" + ("_multi" if multi else ""), (32, 96, 96, 32),
                     (u, R_uvz, PK_uvz, m))


def verify_double_mont(u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m, multi=False):
    return _ext_call("dsv_verify_double_mont" + ("_multi" if multi else ""), (32, 96, 96, 96, 96, 32),
                     (u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m))


def verify_vargen_mont(u, R_uvz, PK_uvz, Gen_uvz, m, multi=False):
    return _ext_call("dsv_verify_vargen_mont" + ("_multi" if multi else ""), (32, 96, 96, 96, 32),
                     (u, R_uvz, PK_uvz, Gen_uvz, m))


_MONT_COLS = {"single": ("dsv_verify_single_mont_cols", (32, 96, 96, 32)),
              "double": ("dsv_verify_double_mont_cols", (32, 96, 96, 96, 96, 32)),
              "vargen": ("dsv_verify_vargen_mont_cols", (32, 96, 96, 96, 32))}


def verify_mont_cols(scheme, cols):
    """dsv_verify_*_mont_cols: the fields of typed objects where they lie.  cols: one uint8 array
    [n, width] per field in the scheme's column order (single: u, R, PK, m; double: u, R, R', PK, PK',
    m; vargen: u, R, PK, Gen, m) — typically VIEWS into arrays of records (numpy structured arrays,
    `records["R"][:, :96]`): only the last axis has to be contiguous, the row stride is passed on."""
    name, widths = _MONT_COLS[scheme]
    if len(cols) != len(widths):
        raise ValueError("%s takes %d columns" % (name, len(widths)))
    n = cols[0].shape[0]
    arr = (_lib.Column * len(cols))()
    for k, (c, w) in enumerate(zip(cols, widths)):
        if c.dtype != np.uint8 or c.ndim != 2 or c.shape != (n, w) or (w > 1 and c.strides[1] != 1) \
                or c.strides[0] < w:
            raise ValueError("column %d: expected uint8 [n, %d] rows with contiguous bytes, got %r / strides %r"
                             % (k, w, c.shape, c.strides))
        arr[k].base = c.ctypes.data
        arr[k].stride = c.strides[0]
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(getattr(_lib.load(), name)(arr, ctypes.c_size_t(n), _p(ok)))
    return ok


def verify_mont_cols_rlc(scheme, cols):
    """dsv_verify_*_mont_cols_rlc: the same columns through the batch fast accept -> (verdicts, accepted)"""
    name, widths = _MONT_COLS[scheme]
    if len(cols) != len(widths):
        raise ValueError("%s takes %d columns" % (name, len(widths)))
    n = cols[0].shape[0]
    arr = (_lib.Column * len(cols))()
    for k, (c, w) in enumerate(zip(cols, widths)):
        if c.dtype != np.uint8 or c.ndim != 2 or c.shape != (n, w) or (w > 1 and c.strides[1] != 1) \
                or c.strides[0] < w:
            raise ValueError("column %d: expected uint8 [n, %d] rows with contiguous bytes, got %r / strides %r"
                             % (k, w, c.shape, c.strides))
        arr[k].base = c.ctypes.data
        arr[k].stride = c.strides[0]
    ok = np.zeros(n, dtype=np.uint8)
    accepted = ctypes.c_int(0)
    _lib.check(getattr(_lib.load(), name + "_rlc")(arr, ctypes.c_size_t(n), _p(ok), ctypes.byref(accepted)))
    return ok, bool(accepted.value)


class MontColsJob:
    """A batch in flight (dsv_verify_*_mont_cols_submit): `wait()` blocks until the verdicts are
    there and returns them.  Keeps the column arrays alive until then."""

    def __init__(self, scheme, cols):
        name, widths = _MONT_COLS[scheme]
        if len(cols) != len(widths):
            raise ValueError("%s takes %d columns" % (name, len(widths)))
        n = cols[0].shape[0]
        arr = (_lib.Column * len(cols))()
        for k, (c, w) in enumerate(zip(cols, widths)):
            if c.dtype != np.uint8 or c.ndim != 2 or c.shape != (n, w) or (w > 1 and c.strides[1] != 1) \
                    or c.strides[0] < w:
                raise ValueError("column %d: expected uint8 [n, %d] rows with contiguous bytes" % (k, w))
            arr[k].base = c.ctypes.data
            arr[k].stride = c.strides[0]
        self._cols = cols
        self._ok = np.zeros(n, dtype=np.uint8)
        self._job = ctypes.c_void_p()
        _lib.check(getattr(_lib.load(), name + "_submit")(arr, ctypes.c_size_t(n), _p(self._ok),
                                                          ctypes.byref(self._job)))

    def __del__(self):
        # a job dropped unwaited still reads the columns and writes the verdicts: wait before they go
        try:
            if getattr(self, "_job", None) is not None and self._job.value is not None:
                self.wait()
        except Exception:  # noqa: BLE001
            pass

    def done(self):
        return self._job.value is None or _lib.load().dsv_job_done(self._job) == 1

    def wait(self):
        if self._job.value is not None:
            job, self._job = self._job, ctypes.c_void_p()
            _lib.check(_lib.load().dsv_job_wait(job))
            self._cols = None
        return self._ok


def submit_mont_cols(scheme, cols):
    """Asynchronous verify_mont_cols: returns a MontColsJob at once; up to max_in_flight() batches
    per device overlap (the second one's ramp runs under the first one's tail)."""
    return MontColsJob(scheme, cols)


def max_in_flight():
    return int(_lib.load().dsv_max_in_flight())


def challenge_single(R, m):
    R, m = _arr(R, 64), _arr(m, 32)
    n = _same_n(R, m)
    c = np.zeros((n, 32), dtype=np.uint8)
    _lib.check(_lib.load().dsv_challenge_single(_p(R), _p(m), ctypes.c_size_t(n), _p(c)))
    return c


def challenge_double(R, Rp, m):
    R, Rp, m = _arr(R, 64), _arr(Rp, 64), _arr(m, 32)
    n = _same_n(R, Rp, m)
    c = np.zeros((n, 32), dtype=np.uint8)
    _lib.check(_lib.load().dsv_challenge_double(_p(R), _p(Rp), _p(m), ctypes.c_size_t(n), _p(c)))
    return c


def sign_single(sk, m, r):
    sk, m, r = _arr(sk, 32), _arr(m, 32), _arr(r, 32)
    n = _same_n(sk, m, r)
    u = np.zeros((n, 32), dtype=np.uint8)
    R = np.zeros((n, 64), dtype=np.uint8)
    _lib.check(_lib.load().dsv_sign_single(_p(sk), _p(m), _p(r), ctypes.c_size_t(n), _p(u), _p(R)))
    return u, R


def sign_double(sk, m, r):
    sk, m, r = _arr(sk, 32), _arr(m, 32), _arr(r, 32)
    n = _same_n(sk, m, r)
    u = np.zeros((n, 32), dtype=np.uint8)
    R = np.zeros((n, 64), dtype=np.uint8)
    Rp = np.zeros((n, 64), dtype=np.uint8)
    _lib.check(_lib.load().dsv_sign_double(_p(sk), _p(m), _p(r), ctypes.c_size_t(n), _p(u), _p(R),
                                           _p(Rp)))
    return u, R, Rp


def sign_vargen(sk, Gen, m, r):
    sk, Gen, m, r = _arr(sk, 32), _arr(Gen, 64), _arr(m, 32), _arr(r, 32)
    n = _same_n(sk, Gen, m, r)
    u = np.zeros((n, 32), dtype=np.uint8)
    R = np.zeros((n, 64), dtype=np.uint8)
    _lib.check(_lib.load().dsv_sign_vargen(_p(sk), _p(Gen), _p(m), _p(r), ctypes.c_size_t(n),
                                           _p(u), _p(R)))
    return u, R


def public_keys(sk, which=0, Gen=None):
    sk = _arr(sk, 32)
    n = sk.shape[0]
    PK = np.zeros((n, 64), dtype=np.uint8)
    g = None
    if Gen is not None:
        g = _arr(Gen, 64)
        _same_n(sk, g)
    _lib.check(_lib.load().dsv_public_keys(_p(sk), ctypes.c_int(which),
                                           _p(g) if g is not None else ctypes.c_void_p(0),
                                           ctypes.c_size_t(n), _p(PK)))
    return PK


def decompress_points(comp):
    """JubJubAffine::from_bytes over [n, 32] compressed points -> ([n, 64] affine, ok[n])."""
    comp = _arr(comp, 32)
    n = comp.shape[0]
    out = np.zeros((n, 64), dtype=np.uint8)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_decompress_points(_p(comp), ctypes.c_size_t(n), _p(out), _p(ok)))
    return out, ok


def compress_points(uv):
    """JubJubAffine::to_bytes: v with bit 255 = lowest bit of u (pure byte shuffling)."""
    uv = _arr(uv, 64)
    out = np.zeros((uv.shape[0], 32), dtype=np.uint8)
    _lib.check(_lib.load().dsv_compress_points(_p(uv), ctypes.c_size_t(uv.shape[0]), _p(out)))
    return out


def verify_single_wire(sig64, pk32, m):
    sig, pk, m = _arr(sig64, 64), _arr(pk32, 32), _arr(m, 32)
    n = _same_n(sig, pk, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_single_wire(_p(sig), _p(pk), _p(m), ctypes.c_size_t(n), _p(ok)))
    return ok


def verify_wire_rlc(scheme, sig, pk, m):
    """dsv_verify_*_wire_rlc: serialized records in host memory through the batch fast accept ->
    (verdicts, accepted)"""
    sw, pw = {"single": (64, 32), "double": (96, 64), "vargen": (64, 64)}[scheme]
    sig, pk, m = _arr(sig, sw), _arr(pk, pw), _arr(m, 32)
    n = _same_n(sig, pk, m)
    ok = np.zeros(n, dtype=np.uint8)
    accepted = ctypes.c_int(0)
    _lib.check(getattr(_lib.load(), "dsv_verify_%s_wire_rlc" % scheme)(
        _p(sig), _p(pk), _p(m), ctypes.c_size_t(n), _p(ok), ctypes.byref(accepted)))
    return ok, bool(accepted.value)


def verify_double_wire(sig96, pk64, m):
    sig, pk, m = _arr(sig96, 96), _arr(pk64, 64), _arr(m, 32)
    n = _same_n(sig, pk, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_double_wire(_p(sig), _p(pk), _p(m), ctypes.c_size_t(n), _p(ok)))
    return ok


def verify_vargen_wire(sig64, pk64, m):
    sig, pk, m = _arr(sig64, 64), _arr(pk64, 64), _arr(m, 32)
    n = _same_n(sig, pk, m)
    ok = np.zeros(n, dtype=np.uint8)
    _lib.check(_lib.load().dsv_verify_vargen_wire(_p(sig), _p(pk), _p(m), ctypes.c_size_t(n), _p(ok)))
    return ok


def stdrng_sign_inputs(seed, n, first_item=0):
    """(sk, m, nonce) of items first_item.. of the reference harness's StdRng(seed) stream."""
    sk = np.zeros((n, 32), dtype=np.uint8)
    m = np.zeros((n, 32), dtype=np.uint8)
    r = np.zeros((n, 32), dtype=np.uint8)
    _lib.check(_lib.load().dsv_stdrng_sign_inputs(ctypes.c_uint64(seed), ctypes.c_size_t(first_item),
                                                  ctypes.c_size_t(n), _p(sk), _p(m), _p(r)))
    return sk, m, r


def stdrng_sign_inputs_dev(seed, sk, m, r, first_item=0, stream=None):
    n, dev = _rows((sk, 32, "sk"), (m, 32, "m"), (r, 32, "r"))
    _lib.check(_lib.load().dsv_stdrng_sign_inputs_dev(
        ctypes.c_uint64(seed), ctypes.c_size_t(first_item), ctypes.c_size_t(n), _tp(sk, 32),
        _tp(m, 32), _tp(r, 32), _stream_ptr(stream, dev)))


def debug_table_entry(which, window, digit):
    out = np.zeros(96, dtype=np.uint8)
    _lib.check(_lib.load().dsv_debug_table_entry(ctypes.c_int(which), ctypes.c_int(window),
                                                 ctypes.c_int(digit), _p(out)))
    return out


def fixed_window_bits():
    return int(_lib.load().dsv_fixed_window_bits())


def debug_lattice3(u, c):
    """(x, y, z) the var-generator kernel uses for (u, c), as Python integers per item."""
    u, c = _arr(u, 32), _arr(c, 32)
    n = _same_n(u, c)
    out = np.zeros((n, 128), dtype=np.uint8)
    _lib.check(_lib.load().dsv_debug_lattice3(_p(u), _p(c), ctypes.c_size_t(n), _p(out)))
    res = []
    for row in out:
        v = [int.from_bytes(row[32 * k:32 * k + 32].tobytes(), "little") for k in range(3)]
        res.append(tuple(-v[k] if row[96 + k] else v[k] for k in range(3)))
    return res


def debug_half_scalars(c):
    """(a, b) the fixed-generator kernels use for c (a >= 0, b signed), as Python integers per item."""
    c = _arr(c, 32)
    n = c.shape[0]
    out = np.zeros((n, 96), dtype=np.uint8)
    _lib.check(_lib.load().dsv_debug_half_scalars(_p(c), ctypes.c_size_t(n), _p(out)))
    res = []
    for row in out:
        a, b = (int.from_bytes(row[32 * k:32 * k + 32].tobytes(), "little") for k in range(2))
        res.append((a, -b if row[64] else b))
    return res


def debug_fq_mul(a, b):
    a, b = _arr(a, 32), _arr(b, 32)
    n = _same_n(a, b)
    out = np.zeros((n, 32), dtype=np.uint8)
    _lib.check(_lib.load().dsv_debug_fq_mul(_p(a), _p(b), ctypes.c_size_t(n), _p(out)))
    return out


# ------------------------------------------------------------------ HBM-resident path (torch)
# Every wrapper checks what the C ABI cannot: dtype, device, contiguity, row width, that all
# arrays of a call have the same number of rows and live on one GPU, and that verdict / workspace
# buffers are large enough — a mismatched caller gets a ValueError, not an out-of-bounds kernel.
def _t(t, width=None, name="tensor"):
    import torch

    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
        raise ValueError("%s: expected a contiguous CUDA tensor" % name)
    if width is not None:
        if t.dtype != torch.uint8 or t.dim() != 2 or t.shape[1] != width:
            raise ValueError("%s: expected uint8 [n, %d], got %s %r" % (name, width, t.dtype, tuple(t.shape)))
    return t


def _rows(*named):
    """named: (tensor, width, name) triples -> n; all on one device, all with n rows"""
    n = named[0][0].shape[0]
    dev = named[0][0].device
    for t, w, name in named:
        _t(t, w, name)
        if t.shape[0] != n:
            raise ValueError("batch arrays disagree on n: %s has %d rows, expected %d" % (name, t.shape[0], n))
        if t.device != dev:
            raise ValueError("%s is on %s, the rest of the batch on %s" % (name, t.device, dev))
    return n, dev


def _bytes_out(t, need, dev, name):
    import torch

    _t(t, None, name)
    if t.dtype != torch.uint8 or t.numel() < need:
        raise ValueError("%s: need a uint8 CUDA tensor of >= %d elements, got %s x %d"
                         % (name, need, t.dtype, t.numel()))
    if t.device != dev:
        raise ValueError("%s is on %s, the batch on %s" % (name, t.device, dev))
    return ctypes.c_void_p(t.data_ptr())


def _tp(t, width):
    return ctypes.c_void_p(_t(t, width).data_ptr())


def workspace_bytes(n):
    return int(_lib.load().dsv_workspace_bytes(ctypes.c_size_t(n)))


def mixed_workspace_bytes(n):
    return int(_lib.load().dsv_mixed_workspace_bytes(ctypes.c_size_t(n)))


def split_scratch_bytes(n):
    return int(_lib.load().dsv_split_scratch_bytes(ctypes.c_size_t(n)))


def ext_workspace_bytes(n):
    return int(_lib.load().dsv_ext_workspace_bytes(ctypes.c_size_t(n)))


def wire_workspace_bytes(n):
    return int(_lib.load().dsv_wire_workspace_bytes(ctypes.c_size_t(n)))


def _stream_ptr(stream, dev=None):
    import torch

    s = stream if stream is not None else torch.cuda.current_stream(dev)
    return ctypes.c_void_p(s.cuda_stream)


def verify_single_dev(u, R, PK, m, ok, workspace, stream=None):
    """Enqueue on `stream` (default: torch's current stream of the batch's device); does not
    synchronise."""
    n, dev = _rows((u, 32, "u"), (R, 64, "R"), (PK, 64, "PK"), (m, 32, "m"))
    _lib.check(_lib.load().dsv_verify_single_dev(
        _tp(u, 32), _tp(R, 64), _tp(PK, 64), _tp(m, 32), ctypes.c_size_t(n),
        _bytes_out(ok, n, dev, "ok"), _bytes_out(workspace, workspace_bytes(n), dev, "workspace"),
        _stream_ptr(stream, dev)))


def rlc_workspace_bytes(n, window_bits=0):
    b = int(_lib.load().dsv_rlc_workspace_bytes(ctypes.c_size_t(n), ctypes.c_int(window_bits)))
    if b == 0:
        raise ValueError("window_bits must be 0 or one of 4, 6, 8, 12, 14, 16")
    return b


def rlc_plan_info(scheme, n, window_bits=0, groups=1):
    """dsv_rlc_plan_info as a dict (no GPU needed)"""
    out = (ctypes.c_uint64 * 24)()
    _lib.check(_lib.load().dsv_rlc_plan_info(ctypes.c_int({"single": 0, "double": 1, "vargen": 2}[scheme]),
                                             ctypes.c_size_t(n), ctypes.c_int(window_bits), ctypes.c_int(groups), out))
    names = ("c", "half", "wpk", "wr", "windows", "nseg", "nseg2", "fine_bits", "kmul", "lpts", "spts", "fixed",
             "entries", "buckets", "tmp0", "tmp1", "coarse_bits", "rows", "row_stride", "bins", "bin_cap", "groups",
             "sub", "bytes")
    return dict(zip(names, [int(x) for x in out]))


def rlc_history(device=0, set_to=-1):
    """dsv_debug_rlc_history: the device's fast-accept history counter (> 0: the next call checks a sample
    and runs in sub-groups); set_to >= 0 overrides it.  Returns the value before the call."""
    r = _lib.load().dsv_debug_rlc_history(ctypes.c_int(device), ctypes.c_int(set_to))
    if r < 0:
        _lib.check(r)
    return r


def rlc_history_long(device=0, set_to=-1):
    """dsv_debug_rlc_history_long: the slow counter behind "guarded" calls; returns the value before the call"""
    r = _lib.load().dsv_debug_rlc_history_long(ctypes.c_int(device), ctypes.c_int(set_to))
    if r < 0:
        _lib.check(r)
    return r


def rlc_subgroups(groups=-1):
    """dsv_debug_rlc_subgroups: force `groups` sub-groups per group (0: automatic); returns the previous setting"""
    return int(_lib.load().dsv_debug_rlc_subgroups(ctypes.c_int(groups)))


def _accepted_arg(accepted_out, dev):
    """the `accepted` argument of the *_rlc_dev calls: None -> a host int (the call waits for the stream and
    returns a bool); a one-element int32 tensor on `dev` or in pinned host memory -> written by the device
    when the stream gets there, the call does not block and returns None"""
    if accepted_out is None:
        box = ctypes.c_int(0)
        return ctypes.byref(box), box
    import torch

    t = accepted_out
    if t.dtype != torch.int32 or t.numel() < 1 or not (t.is_cuda and t.device == dev or (not t.is_cuda and t.is_pinned())):
        raise ValueError("accepted_out: need an int32 tensor on the batch's device or in pinned host memory")
    return ctypes.c_void_p(t.data_ptr()), None


def _rlc(name, cols, ok, workspace, stream, window_bits, accepted_out=None):
    n, dev = _rows(*cols)
    arg, box = _accepted_arg(accepted_out, dev)
    _lib.check(getattr(_lib.load(), name)(
        *[_tp(t, w) for t, w, _ in cols], ctypes.c_size_t(n), _bytes_out(ok, n, dev, "ok"),
        _bytes_out(workspace, rlc_workspace_bytes(n, window_bits), dev, "workspace"),
        _stream_ptr(stream, dev), ctypes.c_int(window_bits), arg))
    return bool(box.value) if box is not None else None


def verify_single_rlc_dev(u, R, PK, m, ok, workspace, stream=None, window_bits=0, accepted_out=None):
    """dsv_verify_single_rlc_dev: the verdict vector of verify_single_dev, through one aggregate test
    per group (or sub-group) when it is valid.  Enqueue-only when `accepted_out` (an int32 tensor on the
    device or in pinned host memory) is given: it receives 1 if every group was accepted by its aggregates
    once the stream gets there.  Without it the call waits for `stream` and returns that as a bool."""
    return _rlc("dsv_verify_single_rlc_dev", ((u, 32, "u"), (R, 64, "R"), (PK, 64, "PK"), (m, 32, "m")),
                ok, workspace, stream, window_bits, accepted_out)


def verify_double_rlc_dev(u, R, Rp, PK, PKp, m, ok, workspace, stream=None, window_bits=0, accepted_out=None):
    return _rlc("dsv_verify_double_rlc_dev", ((u, 32, "u"), (R, 64, "R"), (Rp, 64, "Rp"), (PK, 64, "PK"),
                                               (PKp, 64, "PKp"), (m, 32, "m")), ok, workspace, stream, window_bits,
                accepted_out)


def verify_vargen_rlc_dev(u, R, PK, Gen, m, ok, workspace, stream=None, window_bits=0, accepted_out=None):
    return _rlc("dsv_verify_vargen_rlc_dev", ((u, 32, "u"), (R, 64, "R"), (PK, 64, "PK"), (Gen, 64, "Gen"),
                                               (m, 32, "m")), ok, workspace, stream, window_bits, accepted_out)


def verify_double_dev(u, R, Rp, PK, PKp, m, ok, workspace, stream=None):
    n, dev = _rows((u, 32, "u"), (R, 64, "R"), (Rp, 64, "Rp"), (PK, 64, "PK"), (PKp, 64, "PKp"),
                   (m, 32, "m"))
    _lib.check(_lib.load().dsv_verify_double_dev(
        _tp(u, 32), _tp(R, 64), _tp(Rp, 64), _tp(PK, 64), _tp(PKp, 64), _tp(m, 32),
        ctypes.c_size_t(n), _bytes_out(ok, n, dev, "ok"),
        _bytes_out(workspace, workspace_bytes(n), dev, "workspace"), _stream_ptr(stream, dev)))


def verify_vargen_dev(u, R, PK, Gen, m, ok, workspace, stream=None):
    n, dev = _rows((u, 32, "u"), (R, 64, "R"), (PK, 64, "PK"), (Gen, 64, "Gen"), (m, 32, "m"))
    _lib.check(_lib.load().dsv_verify_vargen_dev(
        _tp(u, 32), _tp(R, 64), _tp(PK, 64), _tp(Gen, 64), _tp(m, 32), ctypes.c_size_t(n),
        _bytes_out(ok, n, dev, "ok"), _bytes_out(workspace, workspace_bytes(n), dev, "workspace"),
        _stream_ptr(stream, dev)))


def verify_single_ext_dev(u, R_uvz, PK_uvz, m, ok, workspace, stream=None):
    n, dev = _rows((u, 32, "u"), (R_uvz, 96, "R_uvz"), (PK_uvz, 96, "PK_uvz"), (m, 32, "m"))
    _lib.check(_lib.load().dsv_verify_single_ext_dev(
        _tp(u, 32), _tp(R_uvz, 96), _tp(PK_uvz, 96), _tp(m, 32), ctypes.c_size_t(n),
        _bytes_out(ok, n, dev, "ok"), _bytes_out(workspace, ext_workspace_bytes(n), dev, "workspace"),
        _stream_ptr(stream, dev)))


def verify_double_ext_dev(u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m, ok, workspace, stream=None):
    n, dev = _rows((u, 32, "u"), (R_uvz, 96, "R_uvz"), (Rp_uvz, 96, "Rp_uvz"), (PK_uvz, 96, "PK_uvz"),
                   (PKp_uvz, 96, "PKp_uvz"), (m, 32, "m"))
    _lib.check(_lib.load().dsv_verify_double_ext_dev(
        _tp(u, 32), _tp(R_uvz, 96), _tp(Rp_uvz, 96), _tp(PK_uvz, 96), _tp(PKp_uvz, 96), _tp(m, 32),
        ctypes.c_size_t(n), _bytes_out(ok, n, dev, "ok"),
        _bytes_out(workspace, ext_workspace_bytes(n), dev, "workspace"), _stream_ptr(stream, dev)))


def verify_vargen_ext_dev(u, R_uvz, PK_uvz, Gen_uvz, m, ok, workspace, stream=None):
    n, dev = _rows((u, 32, "u"), (R_uvz, 96, "R_uvz"), (PK_uvz, 96, "PK_uvz"), (Gen_uvz, 96, "Gen_uvz"),
                   (m, 32, "m"))
    _lib.check(_lib.load().dsv_verify_vargen_ext_dev(
        _tp(u, 32), _tp(R_uvz, 96), _tp(PK_uvz, 96), _tp(Gen_uvz, 96), _tp(m, 32), ctypes.c_size_t(n),
        _bytes_out(ok, n, dev, "ok"), _bytes_out(workspace, ext_workspace_bytes(n), dev, "workspace"),
        _stream_ptr(stream, dev)))


def mont_workspace_bytes(n):
    return int(_lib.load().dsv_mont_workspace_bytes(ctypes.c_size_t(n)))


def _mont_dev(name, widths, arrays, ok, workspace, stream):
    names = ["col%d" % k for k in range(len(arrays))]
    n, dev = _rows(*[(a, w, nm) for a, w, nm in zip(arrays, widths, names)])
    _lib.check(getattr(_lib.load(), name)(
        *([_tp(a, w) for a, w in zip(arrays, widths)] +
          [ctypes.c_size_t(n), _bytes_out(ok, n, dev, "ok"),
           _bytes_out(workspace, mont_workspace_bytes(n), dev, "workspace"), _stream_ptr(stream, dev)])))


def verify_single_mont_dev(u, R_uvz, PK_uvz, m, ok, workspace, stream=None):
    """Montgomery limbs (the Rust types' in-memory form) resident in HBM."""
    _mont_dev("dsv_verify_single_mont_dev", (32, 96, 96, 32), (u, R_uvz, PK_uvz, m), ok, workspace, stream)


def verify_double_mont_dev(u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m, ok, workspace, stream=None):
    _mont_dev("dsv_verify_double_mont_dev", (32, 96, 96, 96, 96, 32),
              (u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m), ok, workspace, stream)


def verify_vargen_mont_dev(u, R_uvz, PK_uvz, Gen_uvz, m, ok, workspace, stream=None):
    _mont_dev("dsv_verify_vargen_mont_dev", (32, 96, 96, 96, 32), (u, R_uvz, PK_uvz, Gen_uvz, m), ok,
              workspace, stream)


def _wire_dev(name, sig, sig_w, pk, pk_w, m, ok, workspace, stream):
    n, dev = _rows((sig, sig_w, "sig"), (pk, pk_w, "pk"), (m, 32, "m"))
    _lib.check(getattr(_lib.load(), name)(
        _tp(sig, sig_w), _tp(pk, pk_w), _tp(m, 32), ctypes.c_size_t(n), _bytes_out(ok, n, dev, "ok"),
        _bytes_out(workspace, wire_workspace_bytes(n), dev, "workspace"), _stream_ptr(stream, dev)))


def verify_single_wire_dev(sig64, pk32, m, ok, workspace, stream=None):
    """Serialized records resident in HBM: Signature (64 B) / PublicKey (32 B) per item."""
    _wire_dev("dsv_verify_single_wire_dev", sig64, 64, pk32, 32, m, ok, workspace, stream)


def verify_double_wire_dev(sig96, pk64, m, ok, workspace, stream=None):
    _wire_dev("dsv_verify_double_wire_dev", sig96, 96, pk64, 64, m, ok, workspace, stream)


def verify_vargen_wire_dev(sig64, pk64, m, ok, workspace, stream=None):
    _wire_dev("dsv_verify_vargen_wire_dev", sig64, 64, pk64, 64, m, ok, workspace, stream)


def wire_rlc_workspace_bytes(n, window_bits=0):
    b = int(_lib.load().dsv_wire_rlc_workspace_bytes(ctypes.c_size_t(n), ctypes.c_int(window_bits)))
    if b == 0:
        raise ValueError("window_bits must be 0 or one of 4, 6, 8, 12, 14, 16")
    return b


_WIRE_WIDTHS = {"single": (64, 32), "double": (96, 64), "vargen": (64, 64)}


def verify_wire_rlc_dev(scheme, sig, pk, m, ok, workspace, stream=None, window_bits=0):
    """dsv_verify_*_wire_rlc_dev: serialized records resident in HBM through the batch fast accept.
    Blocks on `stream`; returns True if the aggregate decided every group."""
    sw, pw = _WIRE_WIDTHS[scheme]
    n, dev = _rows((sig, sw, "sig"), (pk, pw, "pk"), (m, 32, "m"))
    accepted = ctypes.c_int(0)
    _lib.check(getattr(_lib.load(), "dsv_verify_%s_wire_rlc_dev" % scheme)(
        _tp(sig, sw), _tp(pk, pw), _tp(m, 32), ctypes.c_size_t(n), _bytes_out(ok, n, dev, "ok"),
        _bytes_out(workspace, wire_rlc_workspace_bytes(n, window_bits), dev, "workspace"),
        _stream_ptr(stream, dev), ctypes.c_int(window_bits), ctypes.byref(accepted)))
    return bool(accepted.value)


def verify_core_dev(u, c, valid, PK, R, ok, workspace, which=0, accumulate=False, stream=None):
    n, dev = _rows((u, 32, "u"), (c, 32, "c"), (PK, 64, "PK"), (R, 64, "R"))
    _lib.check(_lib.load().dsv_verify_core_dev(
        _tp(u, 32), _tp(c, 32), _bytes_out(valid, n, dev, "valid"), _tp(PK, 64), _tp(R, 64),
        ctypes.c_int(which), ctypes.c_int(1 if accumulate else 0), ctypes.c_size_t(n),
        _bytes_out(ok, n, dev, "ok"), _bytes_out(workspace, workspace_bytes(n), dev, "workspace"),
        _stream_ptr(stream, dev)))


def verify_core_double_dev(u, c, valid, PK, R, PKp, Rp, ok, workspace, stream=None):
    n, dev = _rows((u, 32, "u"), (c, 32, "c"), (PK, 64, "PK"), (R, 64, "R"), (PKp, 64, "PKp"),
                   (Rp, 64, "Rp"))
    _lib.check(_lib.load().dsv_verify_core_double_dev(
        _tp(u, 32), _tp(c, 32), _bytes_out(valid, n, dev, "valid"), _tp(PK, 64), _tp(R, 64),
        _tp(PKp, 64), _tp(Rp, 64), ctypes.c_size_t(n), _bytes_out(ok, n, dev, "ok"),
        _bytes_out(workspace, workspace_bytes(n), dev, "workspace"), _stream_ptr(stream, dev)))


def verify_mixed_dev(kinds, u, R, Rp, PK, PKp, m, n_double, ok, workspace, stream=None):
    """dsv_verify_mixed_dev: kinds uint8 [n] (0 single, 1 double), SoA over all n items."""
    n, dev = _rows((u, 32, "u"), (R, 64, "R"), (Rp, 64, "Rp"), (PK, 64, "PK"), (PKp, 64, "PKp"),
                   (m, 32, "m"))
    _lib.check(_lib.load().dsv_verify_mixed_dev(
        _bytes_out(kinds, n, dev, "kinds"), _tp(u, 32), _tp(R, 64), _tp(Rp, 64), _tp(PK, 64),
        _tp(PKp, 64), _tp(m, 32), ctypes.c_size_t(n), ctypes.c_size_t(int(n_double)),
        _bytes_out(ok, n, dev, "ok"), _bytes_out(workspace, mixed_workspace_bytes(n), dev, "workspace"),
        _stream_ptr(stream, dev)))


def mixed_rlc_workspace_bytes(n):
    return int(_lib.load().dsv_mixed_rlc_workspace_bytes(ctypes.c_size_t(n)))


def verify_mixed_rlc_dev(kinds, u, R, Rp, PK, PKp, m, n_double, ok, workspace, stream=None):
    """dsv_verify_mixed_rlc_dev: the mixed batch with each kind through the batch fast accept; waits for
    `stream` (a host int receives the verdict); returns True if every group of both kinds was decided by its
    aggregates."""
    n, dev = _rows((u, 32, "u"), (R, 64, "R"), (Rp, 64, "Rp"), (PK, 64, "PK"), (PKp, 64, "PKp"),
                   (m, 32, "m"))
    accepted = ctypes.c_int(0)
    _lib.check(_lib.load().dsv_verify_mixed_rlc_dev(
        _bytes_out(kinds, n, dev, "kinds"), _tp(u, 32), _tp(R, 64), _tp(Rp, 64), _tp(PK, 64),
        _tp(PKp, 64), _tp(m, 32), ctypes.c_size_t(n), ctypes.c_size_t(int(n_double)),
        _bytes_out(ok, n, dev, "ok"), _bytes_out(workspace, mixed_rlc_workspace_bytes(n), dev, "workspace"),
        _stream_ptr(stream, dev), ctypes.byref(accepted)))
    return bool(accepted.value)


def _idx(t, need, dev, name):
    import torch

    _t(t, None, name)
    if t.dtype not in (torch.int32, torch.uint32) or t.numel() < need or t.device != dev:
        raise ValueError("%s: need a 32-bit index tensor of >= %d elements on %s" % (name, need, dev))
    return ctypes.c_void_p(t.data_ptr())


def split_kinds_dev(kinds, idx_single, idx_double, scratch, stream=None):
    """Stable split of the index vector by kind; returns nothing — the two counts are the last two
    int32 of `scratch` (view: scratch[-256:-248].view(torch.int32))."""
    n = kinds.numel()
    dev = kinds.device
    _lib.check(_lib.load().dsv_split_kinds_dev(
        _bytes_out(kinds, n, dev, "kinds"), ctypes.c_size_t(n),
        _idx(idx_single, 0, dev, "idx_single"), ctypes.c_size_t(idx_single.numel()),
        _idx(idx_double, 0, dev, "idx_double"), ctypes.c_size_t(idx_double.numel()),
        _bytes_out(scratch, split_scratch_bytes(n), dev, "scratch"), _stream_ptr(stream, dev)))


def split_counts(scratch):
    """The two device-side counts (int32 [2] view: kind 0, kind 1) the last split left in `scratch`."""
    import torch

    return scratch[-256:-248].view(torch.int32)


def _limit_ptr(limit, dev):
    import torch

    if limit is None:
        return ctypes.c_void_p(0)
    if not (isinstance(limit, torch.Tensor) and limit.is_cuda and limit.device == dev
            and limit.dtype in (torch.int32, torch.uint32) and limit.numel() >= 1):
        raise ValueError("limit: need a 32-bit CUDA tensor of one element on %s" % dev)
    return ctypes.c_void_p(limit.data_ptr())


def gather_rows_dev(src, idx, count, dst, limit=None, stream=None):
    """dst[j] = src[idx[j]] for j < min(count, limit[0]); src/dst uint8 [*, row_bytes], row_bytes %
    16 == 0.  `limit`: one-element 32-bit device tensor (e.g. split_counts(scratch)[k:k+1]) —
    entries of idx beyond it are never read; an index >= src.shape[0] is skipped."""
    _t(src, src.shape[1], "src")
    _t(dst, src.shape[1], "dst")
    dev = src.device
    if dst.shape[0] < count or dst.device != dev:
        raise ValueError("dst too small or on another device")
    _lib.check(_lib.load().dsv_gather_rows_dev(
        ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(src.shape[0]), ctypes.c_size_t(src.shape[1]),
        _idx(idx, count, dev, "idx"), ctypes.c_size_t(count), _limit_ptr(limit, dev),
        ctypes.c_void_p(dst.data_ptr()), _stream_ptr(stream, dev)))


def scatter_verdicts_dev(src, idx, count, dst, limit=None, stream=None):
    """dst[idx[j]] = src[j] for j < min(count, limit[0]) (uint8 verdicts back into batch order); an
    index >= dst.numel() is skipped."""
    dev = dst.device
    _lib.check(_lib.load().dsv_scatter_verdicts_dev(
        _bytes_out(src, count, dev, "src"), _idx(idx, count, dev, "idx"), ctypes.c_size_t(count),
        _limit_ptr(limit, dev), _bytes_out(dst, 0, dev, "dst"), ctypes.c_size_t(dst.numel()),
        _stream_ptr(stream, dev)))


def challenge_double_dev(R, Rp, m, c, valid=None, stream=None):
    n, dev = _rows((R, 64, "R"), (Rp, 64, "Rp"), (m, 32, "m"), (c, 32, "c"))
    _lib.check(_lib.load().dsv_challenge_double_dev(
        _tp(R, 64), _tp(Rp, 64), _tp(m, 32), ctypes.c_size_t(n), _tp(c, 32),
        _bytes_out(valid, n, dev, "valid") if valid is not None else ctypes.c_void_p(0),
        _stream_ptr(stream, dev)))


def challenge_single_dev(R, m, c, valid=None, stream=None):
    n, dev = _rows((R, 64, "R"), (m, 32, "m"), (c, 32, "c"))
    _lib.check(_lib.load().dsv_challenge_single_dev(
        _tp(R, 64), _tp(m, 32), ctypes.c_size_t(n), _tp(c, 32),
        _bytes_out(valid, n, dev, "valid") if valid is not None else ctypes.c_void_p(0),
        _stream_ptr(stream, dev)))


def decompress_points_dev(comp, out_uv, ok, in_stride=32, accumulate=False, stream=None):
    """comp: uint8 CUDA tensor holding one 32-byte record every in_stride bytes."""
    n = out_uv.shape[0]
    dev = out_uv.device
    _t(comp, None, "comp")
    if comp.numel() * comp.element_size() < (n - 1) * in_stride + 32 or comp.device != dev:
        raise ValueError("comp holds fewer than n records of stride %d (or is on another device)" % in_stride)
    _lib.check(_lib.load().dsv_decompress_points_dev(
        ctypes.c_void_p(comp.data_ptr()), ctypes.c_size_t(in_stride), ctypes.c_size_t(n),
        _tp(out_uv, 64), _bytes_out(ok, n, dev, "ok"), ctypes.c_int(1 if accumulate else 0),
        _stream_ptr(stream, dev)))


def sign_single_dev(sk, m, r, u, R, stream=None):
    n, dev = _rows((sk, 32, "sk"), (m, 32, "m"), (r, 32, "r"), (u, 32, "u"), (R, 64, "R"))
    _lib.check(_lib.load().dsv_sign_single_dev(
        _tp(sk, 32), _tp(m, 32), _tp(r, 32), ctypes.c_size_t(n), _tp(u, 32), _tp(R, 64),
        _stream_ptr(stream, dev)))


def sign_double_dev(sk, m, r, u, R, Rp, stream=None):
    n, dev = _rows((sk, 32, "sk"), (m, 32, "m"), (r, 32, "r"), (u, 32, "u"), (R, 64, "R"), (Rp, 64, "Rp"))
    _lib.check(_lib.load().dsv_sign_double_dev(
        _tp(sk, 32), _tp(m, 32), _tp(r, 32), ctypes.c_size_t(n), _tp(u, 32), _tp(R, 64),
        _tp(Rp, 64), _stream_ptr(stream, dev)))


def public_keys_dev(sk, which, PK, stream=None):
    n, dev = _rows((sk, 32, "sk"), (PK, 64, "PK"))
    _lib.check(_lib.load().dsv_public_keys_dev(
        _tp(sk, 32), ctypes.c_int(which), ctypes.c_size_t(n), _tp(PK, 64), _stream_ptr(stream, dev)))


def public_keys_vargen_dev(sk, Gen, PK, workspace, stream=None):
    n, dev = _rows((sk, 32, "sk"), (Gen, 64, "Gen"), (PK, 64, "PK"))
    _lib.check(_lib.load().dsv_public_keys_vargen_dev(
        _tp(sk, 32), _tp(Gen, 64), ctypes.c_size_t(n), _tp(PK, 64),
        _bytes_out(workspace, workspace_bytes(n), dev, "workspace"), _stream_ptr(stream, dev)))


def sign_vargen_dev(sk, Gen, m, r, u, R, workspace, stream=None):
    n, dev = _rows((sk, 32, "sk"), (Gen, 64, "Gen"), (m, 32, "m"), (r, 32, "r"), (u, 32, "u"),
                   (R, 64, "R"))
    _lib.check(_lib.load().dsv_sign_vargen_dev(
        _tp(sk, 32), _tp(Gen, 64), _tp(m, 32), _tp(r, 32), ctypes.c_size_t(n), _tp(u, 32),
        _tp(R, 64), _bytes_out(workspace, workspace_bytes(n), dev, "workspace"),
        _stream_ptr(stream, dev)))


def stdrng_vargen_inputs_dev(seed, sk, g, m, r, first_item=0, stream=None):
    n, dev = _rows((sk, 32, "sk"), (g, 32, "g"), (m, 32, "m"), (r, 32, "r"))
    _lib.check(_lib.load().dsv_stdrng_vargen_inputs_dev(
        ctypes.c_uint64(seed), ctypes.c_size_t(first_item), ctypes.c_size_t(n), _tp(sk, 32),
        _tp(g, 32), _tp(m, 32), _tp(r, 32), _stream_ptr(stream, dev)))
