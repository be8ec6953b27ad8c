"""Build libdsv.so (the HIP engine) in-tree with hipcc for gfx950.

    python -m schnorr_amd.build [--force] [-DNAME=VALUE ...]

One hipcc job per translation unit (the host units dsv_*.hip and the kernel units k_*.hip; no
relocatable device code), run in parallel, objects under build/obj/; only units whose sources
changed are recompiled.  No GPU is needed to build (hipcc cross-compiles); the resulting
schnorr_amd/libdsv.so is git-ignored but travels to the GPU box with the working tree.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdsv.so")
UNITS = ["dsv_context.hip", "dsv_device.hip", "dsv_host.hip", "dsv_wire.hip", "dsv_rlc.hip", "dsv_inputs.hip", "k_hash.hip", "k_verify.hip", "k_quad.hip", "k_vargen.hip", "k_misc.hip", "k_rlc.hip"]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra"]


def _deps():
    # every source, header, generator (not __pycache__ etc.: importing the generator in a test must
    # not make the library look stale)
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip", ".py"))]
    deps.append(os.path.join(ROOT, "include", "dsv.h"))
    return deps


def _headers_stamp():
    h = hashlib.sha256()
    for d in _deps():
        if d.endswith(".h"):
            with open(d, "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _deps())


# Test builds of single units (the other units' objects are shared with the default build):
#   lat0: the var-generator kernel without its lattice reduction — every item takes the fallback row
#         (u, c, 1), the branch no hash output can steer a challenge to (lattice3.h)
VARIANTS = {"lat0": {"k_vargen.hip": ["-DDSV_LAT_MAX_BATCHES=0"]}}


def variant_path(name):
    return os.path.join(HERE, "libdsv_%s.so" % name)


def _toolchain_stamp(hipcc):
    """what an object depends on besides its sources: compiler, target, flags (a changed
    FLAGS / ARCH / ROCm version must not link stale objects)"""
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError:
        ver = "?"
    return hashlib.sha256((ver + ARCH + " ".join(FLAGS)).encode()).hexdigest()[:12]


def build(force=False, extra_flags=(), verbose=False, out=None, jobs=None, unit_flags=None):
    """Returns the path of the library.  extra_flags / out: A/B builds (another -D set, another file);
    unit_flags = {unit: [flags]}: a variant in which only those units are compiled differently."""
    out = out or LIB
    if not force and not extra_flags and not unit_flags and out == LIB and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP engine (libdsv.so)")
    const_h = os.path.join(CSRC, "dsv_constants.h")
    gen = os.path.join(CSRC, "gen_constants.py")
    if not os.path.exists(const_h) or os.path.getmtime(gen) > os.path.getmtime(const_h):
        subprocess.check_call([sys.executable, gen])
    unit_flags = unit_flags or {}
    stamp = _headers_stamp() + _toolchain_stamp(hipcc)

    def compile_unit(unit):
        flags = list(extra_flags) + list(unit_flags.get(unit, ()))
        tag = hashlib.sha256(" ".join(flags).encode()).hexdigest()[:8] if flags else "default"
        objdir = os.path.join(ROOT, "build", "obj", tag)
        os.makedirs(objdir, exist_ok=True)
        src = os.path.join(CSRC, unit)
        obj = os.path.join(objdir, unit.replace(".hip", ".o"))
        key = obj + ".key"
        with open(src, "rb") as f:
            want = stamp + hashlib.sha256(f.read()).hexdigest()[:16]
        if not force and os.path.exists(obj) and os.path.exists(key) and open(key).read() == want:
            return obj
        cmd = [hipcc] + FLAGS + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(key, "w") as f:
            f.write(want)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_unit, UNITS))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def build_variant(name, force=False, verbose=False):
    """schnorr_amd/libdsv_<name>.so (git-ignored, travels to the GPU box like libdsv.so)"""
    out = variant_path(name)
    deps = _deps()
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps if os.path.exists(d)):
        return out
    return build(force=force, verbose=verbose, out=out, unit_flags=VARIANTS[name])


def _includes(path, seen):
    """the translation unit and every local header it includes, transitively"""
    import re
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return
    seen.add(path)
    with open(path, errors="replace") as f:
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', f.read(), re.M):
            _includes(os.path.join(os.path.dirname(path), m.group(1)), seen)


def unit_sources_sha256(unit):
    """sha256 over a kernel translation unit and the local headers it includes: what a recorded
    profile of that unit's kernels is valid for (profiles/pmc_latest.json, bench.py roofline.evidence)"""
    seen = set()
    _includes(os.path.join(CSRC, unit), seen)
    h = hashlib.sha256()
    for p in sorted(seen):
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def evidence_hashes():
    """identity of the build a profile was taken from: kernel sources (authoritative: the library is
    rebuilt per box and need not be byte-identical) and the library file itself"""
    out = {"k_verify_sources_sha256": unit_sources_sha256("k_verify.hip"),
           "k_hash_sources_sha256": unit_sources_sha256("k_hash.hip")}
    if os.path.exists(LIB):
        h = hashlib.sha256()
        with open(LIB, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        out["libdsv_sha256"] = h.hexdigest()
    return out


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    names = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--variant=")]
    if names:
        for nm in names:
            print(build_variant(nm, force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, extra_flags=flags, verbose=True))
