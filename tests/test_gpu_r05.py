"""GPU tests of what round 5 added.  Parity is always against the CPU oracle (oracle/).

* dsv_verify_*_mont_cols_submit / dsv_job_wait: batches in flight (each call owns a Pipe, the
  compute lanes are shared) — the verdicts of `PublicKey{,Double,VarGen}::verify`
  (/root/reference/src/keys/public.rs:121-130, 222-244, 401-415) whatever else is in flight.
* blocking host entry points called from several threads at once.
"""
import threading

import numpy as np
import pytest

import mont_cases as C

pytestmark = pytest.mark.gpu
SEEDS = {"single": 51, "double": 52, "vargen": 53}


def _tiled_case(scheme, base, n, seed, period=5):
    """a signed + tampered batch incl. z = 0 / limbs >= modulus items, tiled to n items"""
    cols, want = C.mont_case(scheme, base, seed, period=period)
    reps = -(-n // base)
    tcols = [np.ascontiguousarray(np.tile(c, (reps, 1))[:n]) for c in cols]
    return tcols, np.tile(want, reps)[:n]


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_submit_wait_matches_the_oracle(engine, scheme):
    """one job: a small batch (one chunk on the small-call stream) and one of several chunks"""
    assert engine.max_in_flight() >= 2
    for n in (313, (1 << 16) + (1 << 15) + 77):
        tcols, want = _tiled_case(scheme, 313 if n == 313 else 211, n, SEEDS[scheme])
        views = C.as_records(scheme, tcols)[3]
        job = engine.submit_mont_cols(scheme, views)
        got = job.wait()
        assert job.done()
        assert np.array_equal(got, want), (scheme, n)
        assert 0 < want.sum() < n
        assert np.array_equal(job.wait(), want)          # a second wait returns the same verdicts


def test_batches_in_flight_of_different_schemes_and_sizes(engine):
    """Five jobs submitted back to back — three schemes, sizes from one small chunk to several chunks
    with ragged tails — more than dsv_max_in_flight(): the surplus waits for a pipe inside its driver
    thread.  Every job's verdicts are the oracle's, whatever shared the compute lanes with it."""
    plan = [("single", (1 << 17) + 4099, 61), ("double", (1 << 16) + 333, 62), ("vargen", 1 << 16, 63),
            ("single", 700, 64), ("vargen", (1 << 16) + (1 << 14) + 5, 65)]
    cases = []
    for scheme, n, seed in plan:
        tcols, want = _tiled_case(scheme, 257, n, seed, period=4)
        cases.append((scheme, C.as_records(scheme, tcols)[3], want))
    for threads in (4, 1):
        engine.set_host_threads(threads)
        try:
            jobs = [engine.submit_mont_cols(scheme, views) for scheme, views, _ in cases]
            for (scheme, _, want), job in zip(cases, jobs):
                assert np.array_equal(job.wait(), want), (scheme, len(want), threads)
        finally:
            engine.set_host_threads(0)


def test_blocking_host_calls_from_several_threads(engine):
    """Four threads, each calling a different blocking host entry point (affine bytes, projective bytes,
    limbs, wire records) in a loop: two at a time own a pipe, the others queue; no verdict may change."""
    import oracle_lib as O
    import harness as H
    n = (1 << 16) + 1234
    base = 300
    d = O.keygen_sign_single(base, 71, nthreads=4)
    H.tamper(d)
    want = O.verify_single(d["u"], d["R"], d["PK"], d["m"], nthreads=4)
    reps = -(-n // base)
    t = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    twant = np.tile(want, reps)[:n]
    u, R, PK, m = t(d["u"]), t(d["R"]), t(d["PK"]), t(d["m"])
    mcols, mwant = _tiled_case("single", 211, n, 72)
    sig = np.ascontiguousarray(np.concatenate([u, engine.compress_points(R)], axis=1))
    pk = engine.compress_points(PK)
    wire_want = engine.verify_single_wire(sig, pk, m)    # (decoding vs the oracle: tests/test_gpu_parity.py)
    assert np.array_equal(wire_want, twant)
    errors = []

    def worker(fn, expect, label):
        try:
            for _ in range(4):
                if not np.array_equal(fn(), expect):
                    errors.append(label + ": verdicts differ")
        except Exception as e:  # noqa: BLE001
            errors.append("%s: %r" % (label, e))

    th = [threading.Thread(target=worker, args=a) for a in (
        (lambda: engine.verify_single(u, R, PK, m), twant, "affine"),
        (lambda: engine.verify_single_mont(*mcols), mwant, "limbs"),
        (lambda: engine.verify_single_wire(sig, pk, m), twant, "wire"),
        (lambda: engine.verify_single(u[:900], R[:900], PK[:900], m[:900]), twant[:900], "small"))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors


def test_submit_validates_like_the_blocking_form(engine):
    from schnorr_amd import _lib
    import ctypes
    cols, _ = C.mont_case("single", 8, 5, plant=False)
    arr = (_lib.Column * 4)()
    for k, c in enumerate(cols):
        arr[k].base, arr[k].stride = c.ctypes.data, c.strides[0]
    arr[2].stride = 64
    ok = np.zeros(8, np.uint8)
    job = ctypes.c_void_p(1)
    L = _lib.load()
    rc = L.dsv_verify_single_mont_cols_submit(arr, ctypes.c_size_t(8), ctypes.c_void_p(ok.ctypes.data), ctypes.byref(job))
    assert rc == -2 and job.value is None and b"stride" in L.dsv_last_error()
    arr[2].stride = cols[2].strides[0]
    rc = L.dsv_verify_single_mont_cols_submit(arr, ctypes.c_size_t(0), None, ctypes.byref(job))   # empty batch
    assert rc == 0 and job.value is not None
    assert L.dsv_job_wait(job) == 0
    assert L.dsv_job_wait(None) == -2


def test_shutdown_waits_for_the_calls_in_flight(engine):
    """dsv_shutdown while three jobs are in flight (two own a pipe, one waits for one): it returns only
    after all of them have delivered the oracle's verdicts; calls after it fail loudly; dsv_init brings the
    engine back."""
    from schnorr_amd import _lib
    cases = []
    for scheme, n, seed in (("single", (1 << 17) + 99, 81), ("vargen", 1 << 16, 82), ("double", (1 << 16) + 2000, 83)):
        tcols, want = _tiled_case(scheme, 199, n, seed, period=4)
        cases.append((scheme, C.as_records(scheme, tcols)[3], want))
    jobs = [engine.submit_mont_cols(scheme, views) for scheme, views, _ in cases]
    done = {}
    th = threading.Thread(target=lambda: done.setdefault("rc", engine.shutdown()))
    th.start()
    th.join(timeout=120)
    assert not th.is_alive(), "dsv_shutdown did not return"
    try:
        for (scheme, _, want), job in zip(cases, jobs):
            assert job.done()                                  # shutdown has waited for every one of them
            assert np.array_equal(job.wait(), want), scheme
        with pytest.raises(_lib.DsvError):
            engine.verify_mont_cols(cases[0][0], cases[0][1])  # not initialised any more
    finally:
        engine.init(0)
    assert np.array_equal(engine.verify_mont_cols(cases[0][0], cases[0][1]), cases[0][2])
