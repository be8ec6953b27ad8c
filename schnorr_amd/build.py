"""Build libdsv.so (the HIP engine) in-tree with hipcc for gfx950.

    python -m schnorr_amd.build [--force]

No GPU is needed to build (hipcc cross-compiles); the resulting schnorr_amd/libdsv.so is
git-ignored but travels to the GPU box with the working tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdsv.so")
SOURCES = ["dsv.hip"]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    # every source, header, generator (not __pycache__ etc.: importing the generator in a test must
    # not make the library look stale)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".py"))]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "dsv.h"))
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, extra_flags=(), verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP engine (libdsv.so)")
    const_h = os.path.join(CSRC, "dsv_constants.h")
    gen = os.path.join(CSRC, "gen_constants.py")
    if not os.path.exists(const_h) or os.path.getmtime(gen) > os.path.getmtime(const_h):
        subprocess.check_call([sys.executable, gen])
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wextra", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    cmd += list(extra_flags)
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
