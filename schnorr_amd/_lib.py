"""ctypes binding of libdsv.so (include/dsv.h).  There is NO CPU fallback: if the library is
missing or a call fails, an exception is raised."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSV_LIB_PATH") or os.path.join(HERE, "libdsv.so")  # override: A/B builds

# every extern "C" symbol include/dsv.h declares
SYMBOLS = [
    "dsv_init", "dsv_shutdown", "dsv_version", "dsv_last_error", "dsv_device_count",
    "dsv_verify_single", "dsv_verify_double", "dsv_verify_vargen", "dsv_verify_single_ext",
    "dsv_workspace_bytes", "dsv_verify_single_dev", "dsv_verify_double_dev",
    "dsv_verify_vargen_dev", "dsv_verify_core_dev", "dsv_challenge_single", "dsv_challenge_double",
    "dsv_challenge_single_dev", "dsv_challenge_double_dev", "dsv_sign_single", "dsv_sign_double",
    "dsv_sign_vargen", "dsv_public_keys", "dsv_sign_single_dev", "dsv_sign_double_dev",
    "dsv_public_keys_dev", "dsv_compress_points", "dsv_decompress_points", "dsv_decompress_points_dev",
    "dsv_verify_single_wire", "dsv_verify_double_wire", "dsv_verify_vargen_wire",
    "dsv_stdrng_sign_inputs", "dsv_stdrng_sign_inputs_dev",
    "dsv_debug_table_entry", "dsv_fixed_window_bits", "dsv_debug_fq_mul",
    # r02: several devices per process, fused double kernel, mixed batches, var-generator inputs
    "dsv_shutdown_device", "dsv_set_device", "dsv_get_device", "dsv_initialized_devices",
    "dsv_verify_single_multi", "dsv_verify_double_multi", "dsv_verify_vargen_multi",
    "dsv_verify_core_double_dev", "dsv_mixed_workspace_bytes", "dsv_verify_mixed_dev",
    "dsv_split_scratch_bytes", "dsv_split_kinds_dev", "dsv_gather_rows_dev",
    "dsv_scatter_verdicts_dev", "dsv_public_keys_vargen_dev", "dsv_sign_vargen_dev",
    "dsv_stdrng_vargen_inputs_dev",
    # r03: projective inputs for every scheme (host / multi-device / device pointers), wire records
    # in device memory, one-call initialisation
    "dsv_init_visible", "dsv_to_hash_inputs", "dsv_debug_lattice3", "dsv_debug_half_scalars",
    "dsv_verify_double_ext", "dsv_verify_vargen_ext",
    "dsv_verify_single_ext_multi", "dsv_verify_double_ext_multi", "dsv_verify_vargen_ext_multi",
    "dsv_ext_workspace_bytes", "dsv_verify_single_ext_dev", "dsv_verify_double_ext_dev",
    "dsv_verify_vargen_ext_dev",
    "dsv_wire_workspace_bytes", "dsv_verify_single_wire_dev", "dsv_verify_double_wire_dev",
    "dsv_verify_vargen_wire_dev",
    # r04: the reference's in-memory representation (Montgomery limbs), dense arrays / strided
    # columns of typed objects / device pointers
    "dsv_verify_single_mont", "dsv_verify_double_mont", "dsv_verify_vargen_mont",
    "dsv_verify_single_mont_multi", "dsv_verify_double_mont_multi", "dsv_verify_vargen_mont_multi",
    "dsv_verify_single_mont_cols", "dsv_verify_double_mont_cols", "dsv_verify_vargen_mont_cols",
    "dsv_mont_workspace_bytes", "dsv_verify_single_mont_dev", "dsv_verify_double_mont_dev",
    "dsv_verify_vargen_mont_dev", "dsv_set_host_threads",
    # r05: asynchronous form of the column entry points (two batches in flight per device)
    "dsv_verify_single_mont_cols_submit", "dsv_verify_double_mont_cols_submit",
    "dsv_verify_vargen_mont_cols_submit", "dsv_job_wait", "dsv_job_done", "dsv_max_in_flight",
    # r05: random-linear-combination fast accept in front of the per-signature kernels
    "dsv_rlc_workspace_bytes", "dsv_verify_single_rlc_dev", "dsv_verify_double_rlc_dev",
    "dsv_verify_vargen_rlc_dev", "dsv_rlc_plan_info", "dsv_debug_rlc_history", "dsv_debug_rlc_history_long", "dsv_debug_rlc_subgroups", "dsv_device_numa", "dsv_debug_numa_lookup",
    "dsv_verify_single_mont_cols_rlc", "dsv_verify_double_mont_cols_rlc", "dsv_verify_vargen_mont_cols_rlc",
    "dsv_wire_rlc_workspace_bytes", "dsv_verify_single_wire_rlc_dev", "dsv_verify_double_wire_rlc_dev",
    "dsv_verify_vargen_wire_rlc_dev",
    "dsv_verify_single_wire_rlc", "dsv_verify_double_wire_rlc", "dsv_verify_vargen_wire_rlc",
    "dsv_mixed_rlc_workspace_bytes", "dsv_verify_mixed_rlc_dev",
]
_SIZE_T_FUNCS = ("dsv_workspace_bytes", "dsv_mixed_workspace_bytes", "dsv_split_scratch_bytes",
                 "dsv_ext_workspace_bytes", "dsv_wire_workspace_bytes", "dsv_mont_workspace_bytes",
                 "dsv_rlc_workspace_bytes", "dsv_wire_rlc_workspace_bytes", "dsv_mixed_rlc_workspace_bytes")


class Column(ctypes.Structure):
    """dsv_column: one field of n typed objects, item i at base + i * stride"""
    _fields_ = [("base", ctypes.c_void_p), ("stride", ctypes.c_size_t)]


class DsvError(RuntimeError):
    pass


_lib = None


def load():
    """Load libdsv.so (building it first if hipcc is available and it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build as _build  # raises if hipcc is absent

        _build.build()
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 and libdsv.so is linked
    # against the system one.  If libdsv (and with it the system runtime) is loaded first, a later
    # `import torch` + CUDA init in the same process fails with "No HIP GPUs are available"; in the
    # other order the dynamic loader resolves libdsv's dependency to the copy torch already
    # mapped.  The device-tensor entry points of this package need torch anyway, so load it first
    # when it is installed.  (Pure C / Rust callers never see this: they have one runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise DsvError("cannot load the HIP engine %s: %s" % (LIB_PATH, e))
    L.dsv_version.restype = ctypes.c_char_p
    L.dsv_last_error.restype = ctypes.c_char_p
    for name in _SIZE_T_FUNCS:
        getattr(L, name).restype = ctypes.c_size_t
        getattr(L, name).argtypes = [ctypes.c_size_t] + ([ctypes.c_int] if name in ("dsv_rlc_workspace_bytes", "dsv_wire_rlc_workspace_bytes") else [])
    for name in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
        if name not in ("dsv_version", "dsv_last_error") + _SIZE_T_FUNCS:
            fn.restype = ctypes.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise DsvError("dsv error %d: %s" % (rc, load().dsv_last_error().decode()))
