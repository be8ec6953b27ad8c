"""CPU tests of the drop-in boundary: libdsv.so loads, exports exactly what include/dsv.h
declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from schnorr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dsv.h")
NO_GPU = not torch.cuda.is_available()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_all_exported():
    L = _lib.load()
    decl = declared_symbols()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(L, name), "include/dsv.h declares %s but libdsv.so does not export it" % name
    assert sorted(_lib.SYMBOLS) == decl
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (dsv_[a-z0-9_]+)", nm))
    assert exported == set(decl), exported ^ set(decl)


def test_engine_contains_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_version_and_uninitialised_calls_fail_loudly():
    L = _lib.load()
    assert b"gfx950" in L.dsv_version()
    buf = (ctypes.c_uint8 * 64)()
    rc = L.dsv_verify_single(buf, buf, buf, buf, ctypes.c_size_t(1), buf)
    assert rc == -1  # DSV_ERR_NOT_INITIALIZED
    assert b"dsv_init" in L.dsv_last_error()
    with pytest.raises(_lib.DsvError):
        _lib.check(rc)
    assert L.dsv_workspace_bytes(ctypes.c_size_t(1 << 20)) >= (1 << 20) * 33


def test_host_thread_setting_needs_no_device():
    """dsv_set_host_threads: 1..16 sets the copy pool of the host entry points, 0 restores the default
    ($DSV_HOST_THREADS, else 4, at most the machine's hardware threads); the value in force is returned."""
    L = _lib.load()
    default = L.dsv_set_host_threads(0)
    assert 1 <= default <= 16
    assert L.dsv_set_host_threads(1) == 1
    assert L.dsv_set_host_threads(1000) <= 16
    assert L.dsv_set_host_threads(-3) == default and L.dsv_set_host_threads(0) == default


@pytest.mark.skipif(not NO_GPU, reason="only meaningful on a box without a GPU")
def test_no_gpu_means_error_not_fallback():
    L = _lib.load()
    assert L.dsv_device_count() == 0
    assert L.dsv_init(0) == -4  # DSV_ERR_NO_DEVICE
    from schnorr_amd import engine as E
    with pytest.raises(_lib.DsvError):
        E.init(0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under schnorr_amd/ or include/ may reference it."""
    for base in ("schnorr_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", ".rs")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "oracle_lib" not in text and "schnorr_oracle" not in text and \
                        "libschnorr_oracle" not in text, os.path.join(dirpath, f)
