"""The host pipeline's synchronisation code (schnorr_amd/csrc/host_sync.h: copy-thread pool, pipe leases,
compute turns, chunk plan) compiled WITHOUT HIP and run under ThreadSanitizer: the part of a verify call
where a data race would hide, exercised without a GPU (GPU sanitizers are not available on this pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_sync.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_sync")


def test_host_sync_under_thread_sanitizer():
    base = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-Wall", "-Wextra", "-o", EXE, SRC]
    tsan = subprocess.run(base + ["-fsanitize=thread"], capture_output=True, text=True)
    sanitized = tsan.returncode == 0
    if not sanitized:  # no libtsan on this machine: the functional checks still run
        subprocess.check_call(base)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (sanitized, out.stdout[-500:], out.stderr[-3000:])
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
