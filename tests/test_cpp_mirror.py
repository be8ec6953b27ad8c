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


def test_cpp_mirror_host_arithmetic_matches_python_integers():
    """BlsScalar / JubJubScalar of the mirror hold what the reference's types hold — four u64
    Montgomery limbs, R = 2^256 — and their host arithmetic (to_bytes = a Montgomery reduction,
    from_bytes_wide = lo R^2 + hi R^3, *, +, -, invert) is checked line by line against Python
    integers.  Runs without a GPU: the program makes no engine call."""
    import pymodel as M
    from schnorr_amd import _lib
    _lib.load()
    exe = os.path.join(ROOT, "tests", "cpp", "field_vectors")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "field_vectors.cpp"),
                           "-L", os.path.join(ROOT, "schnorr_amd"), "-ldsv",
                           "-Wl,-rpath," + os.path.join(ROOT, "schnorr_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split("\n")
    le = lambda h: int.from_bytes(bytes.fromhex(h), "little")
    seen = {"fq": 0, "fr": 0}
    for line in out:
        if not line.strip():
            continue
        name, a, b, ab, apb, amb, na, ia, w, fw, limbs = line.split()
        p = M.Q if name == "fq" else M.R_ORDER
        a, b = le(a), le(b)
        assert a < p and b < p
        assert le(ab) == a * b % p and le(apb) == (a + b) % p and le(amb) == (a - b) % p
        assert le(na) == -a % p and le(ia) == pow(a, -1, p)
        assert le(fw) == le(w) % p
        assert le(limbs) == (a << 256) % p          # the in-memory form
        seen[name] += 1
    assert seen == {"fq": 40, "fr": 40}


@pytest.mark.gpu
def test_cpp_mirror_reference_tests_pass_on_gpu():
    _compile()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok:")
