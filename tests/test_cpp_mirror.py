"""The C++ host mirror of the reference's key/signature surface (include/dusk_schnorr.hpp):
compile the C++ counterpart of tests/schnorr.rs, schnorr_double.rs, schnorr_var_generator.rs
against libdsv.so and run it on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_schnorr.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_schnorr")


def _compile():
    from schnorr_amd import _lib
    _lib.load()
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", EXE, SRC,
           "-L", os.path.join(ROOT, "schnorr_amd"), "-ldsv",
           "-Wl,-rpath," + os.path.join(ROOT, "schnorr_amd")]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles():
    _compile()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_mirror_reference_tests_pass_on_gpu():
    _compile()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:")
