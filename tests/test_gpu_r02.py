"""GPU tests of what round 2 added: fused double-signature kernel, device-side split of mixed
batches, several callers / devices, the var-generator input stream, input validation, and the
self-spawning multi-rank bench.  Parity is always against the CPU oracle."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import harness as H
import oracle_lib as O
import pymodel as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def test_double_batch_at_config_size_with_oracle_sample(engine):
    """BASELINE.json configs[2] at its full size: 2^20 double signatures through the fused
    two-equation kernel; expected pattern everywhere, CPU oracle on a strided sample."""
    import torch
    from schnorr_amd import workload as W
    n = 1 << 20
    b = W.gen_double(n, seed=77)
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    engine.verify_double_dev(b["u"], b["R"], b["Rp"], b["PK"], b["PKp"], b["m"], ok, ws)
    torch.cuda.synchronize()
    assert torch.equal(ok, b["expected"])
    assert 0 < int(ok.sum()) < n
    idx = torch.arange(0, n, 1021, device="cuda:0")[:512]
    sub = {k: b[k][idx].cpu().numpy() for k in ("u", "R", "Rp", "PK", "PKp", "m")}
    want = O.verify_double(sub["u"], sub["R"], sub["Rp"], sub["PK"], sub["PKp"], sub["m"], nthreads=8)
    assert np.array_equal(want, ok[idx].cpu().numpy())
    assert 0 < want.sum() < len(want)


def test_fused_double_kernel_equals_two_single_equation_passes(engine):
    """k_verify_fixed_half<false,2> against the r01 formulation (the single-equation kernel twice,
    AND-ing into ok) and against the oracle, on a batch where either half alone can be wrong."""
    import torch
    n = 2048 + 37
    d = O.keygen_sign_double(n, 31, nthreads=8)
    H.tamper(d, period=5)
    for i in range(3, n, 11):                       # break ONLY the primed half
        d["PKp"][i] = d["PKp"][(i + 1) % n]
    for i in range(7, n, 13):                       # break ONLY the plain half
        d["PK"][i] = d["PK"][(i + 1) % n]
    want = O.verify_double(d["u"], d["R"], d["Rp"], d["PK"], d["PKp"], d["m"], nthreads=8)
    t = {k: _dev(v) for k, v in d.items() if k in ("u", "R", "Rp", "PK", "PKp", "m")}
    c = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
    valid = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    engine.challenge_double_dev(t["R"], t["Rp"], t["m"], c, valid)
    fused = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    engine.verify_core_double_dev(t["u"], c, valid, t["PK"], t["R"], t["PKp"], t["Rp"], fused, ws)
    two = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    engine.verify_core_dev(t["u"], c, valid, t["PK"], t["R"], two, ws, which=0)
    engine.verify_core_dev(t["u"], c, valid, t["PKp"], t["Rp"], two, ws, which=1, accumulate=True)
    torch.cuda.synchronize()
    assert np.array_equal(fused.cpu().numpy(), want)
    assert torch.equal(fused, two)
    assert 0 < want.sum() < n


@pytest.mark.parametrize("n", [2, 5, 63, 64, 65, (1 << 14) - 1, 1 << 14, (1 << 14) + 1])
def test_four_lane_kernel_boundary_sizes(engine, n):
    """Batches of <= 2^14 items take k_verify_fixed_half_quad (four lanes per signature, quad29.h),
    larger ones the one-lane kernel: ragged sizes around the workgroup (64 items) and around the
    switch-over, single and double, against the expected pattern and an oracle sample."""
    import torch
    from schnorr_amd import workload as W
    ws = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    for kind in ("single", "double"):
        b = (W.gen_single if kind == "single" else W.gen_double)(n, seed=900 + n % 97)
        ok = torch.full((n,), 9, dtype=torch.uint8, device="cuda:0")
        if kind == "single":
            engine.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
        else:
            engine.verify_double_dev(b["u"], b["R"], b["Rp"], b["PK"], b["PKp"], b["m"], ok, ws)
        torch.cuda.synchronize()
        assert torch.equal(ok, b["expected"]), (kind, n)
        k = min(n, 96)
        cols = ("u", "R", "PK", "m") if kind == "single" else ("u", "R", "Rp", "PK", "PKp", "m")
        sub = [b[c][:k].cpu().numpy() for c in cols]
        want = (O.verify_single if kind == "single" else O.verify_double)(*sub, nthreads=8)
        assert np.array_equal(want, ok[:k].cpu().numpy()), (kind, n)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4095, 4096, 4097, 100003])
def test_split_kinds_matches_numpy(engine, n):
    import torch
    rng = np.random.default_rng(n)
    kinds = rng.integers(0, 2, size=n, dtype=np.uint8)
    if n > 20:
        kinds[5] = 2                                # an invalid kind lands in neither list
        kinds[n - 3] = 255
    want_s, want_d = np.nonzero(kinds == 0)[0], np.nonzero(kinds == 1)[0]
    k = _dev(kinds)
    idx_s = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    idx_d = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    scratch = torch.empty(engine.split_scratch_bytes(n), dtype=torch.uint8, device="cuda:0")
    engine.split_kinds_dev(k, idx_s, idx_d, scratch)
    torch.cuda.synchronize()
    counts = scratch[-256:-248].view(torch.int32).cpu().numpy()
    assert (int(counts[0]), int(counts[1])) == (len(want_s), len(want_d))
    assert np.array_equal(idx_s[:len(want_s)].cpu().numpy(), want_s)
    assert np.array_equal(idx_d[:len(want_d)].cpu().numpy(), want_d)
    assert int((idx_s[len(want_s):] != -1).sum()) == 0   # nothing written past the counts


def _mixed_arrays(n, seed):
    rng = np.random.default_rng(seed)
    kinds = rng.integers(0, 2, size=n, dtype=np.uint8)
    si, di = np.nonzero(kinds == 0)[0], np.nonzero(kinds == 1)[0]
    ds = O.keygen_sign_single(len(si), seed, nthreads=8)
    dd = O.keygen_sign_double(len(di), seed + 1, nthreads=8)
    H.tamper(ds, period=5)
    H.tamper(dd, period=7)
    cols = {k: np.zeros((n, w), np.uint8) for k, w in (("u", 32), ("R", 64), ("Rp", 64), ("PK", 64),
                                                       ("PKp", 64), ("m", 32))}
    for k in ("u", "R", "PK", "m"):
        cols[k][si] = ds[k]
    for k in cols:
        cols[k][di] = dd[k]
    want = np.zeros(n, np.uint8)
    want[si] = O.verify_single(ds["u"], ds["R"], ds["PK"], ds["m"], nthreads=8)
    want[di] = O.verify_double(dd["u"], dd["R"], dd["Rp"], dd["PK"], dd["PKp"], dd["m"], nthreads=8)
    return kinds, cols, want, len(di)


def test_mixed_batch_split_on_the_device(engine):
    """BASELINE.json configs[4] path on one GPU: arbitrary interleaving of single and double
    signatures in one structure of arrays, dsv_verify_mixed_dev (device-side split, row gathers,
    per-kind kernels, scatter back) against the oracle; an unknown kind gives verdict 0; a wrong
    n_double gives an all-zero vector instead of an overrun."""
    import torch
    n = 5000 + 7
    kinds, cols, want, nd = _mixed_arrays(n, 11)
    assert 0 < want.sum() < n
    t = {k: _dev(v) for k, v in cols.items()}
    ok = torch.full((n,), 7, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(engine.mixed_workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    engine.verify_mixed_dev(_dev(kinds), t["u"], t["R"], t["Rp"], t["PK"], t["PKp"], t["m"], nd, ok, ws)
    torch.cuda.synchronize()
    assert np.array_equal(ok.cpu().numpy(), want)
    # all single / all double
    for fill in (0, 1):
        k2 = np.full(n, fill, np.uint8)
        engine.verify_mixed_dev(_dev(k2), t["u"], t["R"], t["Rp"], t["PK"], t["PKp"], t["m"],
                                n * fill, ok, ws)
        torch.cuda.synchronize()
        w2 = np.where(kinds == fill, want, 0)
        assert np.array_equal(ok.cpu().numpy()[kinds == fill], w2[kinds == fill])
    # an item of unknown kind: its verdict is 0, the others are unaffected
    k3 = kinds.copy()
    victim = int(np.nonzero((kinds == 0) & (want == 1))[0][0])
    k3[victim] = 9
    engine.verify_mixed_dev(_dev(k3), t["u"], t["R"], t["Rp"], t["PK"], t["PKp"], t["m"], nd, ok, ws)
    torch.cuda.synchronize()
    got = ok.cpu().numpy()
    assert got.sum() == 0                           # count of kind 0 no longer matches n - n_double
    # declared count off by one: no usable verdicts, no fault
    engine.verify_mixed_dev(_dev(kinds), t["u"], t["R"], t["Rp"], t["PK"], t["PKp"], t["m"], nd - 1, ok, ws)
    torch.cuda.synchronize()
    assert int(ok.sum()) == 0


def test_vargen_input_stream_matches_restatement(engine):
    """dsv_stdrng_vargen_inputs_dev: four from_bytes_wide draws per item (sk, generator scalar,
    message, nonce — src/keys/secret.rs:371-373, tests/schnorr_var_generator.rs:16-22) against
    tests/refrng.py, incl. an offset into the stream; the workload built from it verifies."""
    import torch
    import refrng
    from schnorr_amd import workload as W
    for seed, first, n in ((2321, 0, 200), (777, 500, 90)):
        new = lambda: torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
        sk, g, m, r = new(), new(), new(), new()
        engine.stdrng_vargen_inputs_dev(seed, sk, g, m, r, first_item=first)
        torch.cuda.synchronize()
        sk, g, m, r = (x.cpu().numpy() for x in (sk, g, m, r))
        rng = refrng.StdRng(seed)
        rng.fill_bytes(256 * first)
        for i in range(n):
            a = int.from_bytes(rng.fill_bytes(64), "little") % M.R_ORDER
            b = int.from_bytes(rng.fill_bytes(64), "little") % M.R_ORDER
            c = int.from_bytes(rng.fill_bytes(64), "little") % M.Q
            d = int.from_bytes(rng.fill_bytes(64), "little") % M.R_ORDER
            assert (M.from_le(sk[i]), M.from_le(g[i]), M.from_le(m[i]), M.from_le(r[i])) == (a, b, c, d)
    bv = W.gen_vargen(600, seed=777)
    h = {k: bv[k].cpu().numpy() for k in ("u", "R", "PK", "Gen", "m")}
    want = O.verify_vargen(h["u"], h["R"], h["PK"], h["Gen"], h["m"], nthreads=8)
    assert np.array_equal(want, bv["expected"].cpu().numpy()) and 0 < want.sum() < 600
    assert np.array_equal(engine.verify_vargen(h["u"], h["R"], h["PK"], h["Gen"], h["m"]), want)


def test_signing_rejects_non_canonical_scalars(engine):
    """sk or nonce >= r is an error on the host entry points and poisons the item's
    outputs (0xff.., never a valid encoding) on the device ones — no silent truncation."""
    import torch
    from schnorr_amd import _lib
    n = 8
    sk, m, r = engine.stdrng_sign_inputs(5, n)
    bad_sk = sk.copy()
    bad_sk[3] = np.frombuffer(M.le32(M.R_ORDER), dtype=np.uint8)          # exactly r
    bad_r = r.copy()
    bad_r[5] = 0xFF
    for args in ((bad_sk, m, r), (sk, m, bad_r)):
        with pytest.raises(_lib.DsvError, match="canonical"):
            engine.sign_single(*args)
        with pytest.raises(_lib.DsvError, match="canonical"):
            engine.sign_double(*args)
    with pytest.raises(_lib.DsvError, match="canonical"):
        engine.public_keys(bad_sk)
    u = torch.zeros((n, 32), dtype=torch.uint8, device="cuda:0")
    R = torch.zeros((n, 64), dtype=torch.uint8, device="cuda:0")
    PK = torch.zeros((n, 64), dtype=torch.uint8, device="cuda:0")
    engine.sign_single_dev(_dev(bad_sk), _dev(m), _dev(bad_r), u, R)
    engine.public_keys_dev(_dev(bad_sk), 0, PK)
    torch.cuda.synchronize()
    u, R, PK = u.cpu().numpy(), R.cpu().numpy(), PK.cpu().numpy()
    assert (u[3] == 0xFF).all() and (u[5] == 0xFF).all() and (R[5] == 0xFF).all() and (PK[3] == 0xFF).all()
    good_u, good_R = engine.sign_single(sk, m, r)
    for i in (0, 1, 2, 4, 6, 7):
        assert np.array_equal(u[i], good_u[i]) and np.array_equal(R[i], good_R[i])
    # poisoned outputs can never verify
    ok = engine.verify_single(u, R, engine.public_keys(sk), m)
    assert list(ok) == [1, 1, 1, 0, 1, 0, 1, 1]


def test_device_wrappers_validate_their_tensors(engine):
    """row-count, dtype, device, size mismatches raise instead of reaching a kernel."""
    import torch
    n = 64
    z = lambda rows, w: torch.zeros((rows, w), dtype=torch.uint8, device="cuda:0")
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(ValueError, match="disagree"):
        engine.verify_single_dev(z(n, 32), z(n - 1, 64), z(n, 64), z(n, 32), ok, ws)
    with pytest.raises(ValueError, match="workspace"):
        engine.verify_single_dev(z(n, 32), z(n, 64), z(n, 64), z(n, 32), ok, ws[:1000])
    with pytest.raises(ValueError, match="ok"):
        engine.verify_single_dev(z(n, 32), z(n, 64), z(n, 64), z(n, 32), ok[:10], ws)
    with pytest.raises(ValueError, match="ok"):
        engine.verify_single_dev(z(n, 32), z(n, 64), z(n, 64), z(n, 32), torch.zeros(n, dtype=torch.uint8), ws)
    with pytest.raises(ValueError, match="ok"):
        engine.verify_single_dev(z(n, 32), z(n, 64), z(n, 64), z(n, 32),
                                 torch.zeros(n, dtype=torch.int32, device="cuda:0"), ws)
    with pytest.raises(ValueError):
        engine.verify_double_dev(z(n, 32), z(n, 64), z(n, 32), z(n, 64), z(n, 64), z(n, 32), ok, ws)


def test_multi_device_entry_points_on_one_gpu(engine):
    """dsv_verify_*_multi: with one initialised device it equals the plain host entry point;
    DSV_MULTI_SHARDS forces the multi-shard path (contiguous shards, one host thread each) so the
    sharding, the worker threads and the error propagation are exercised on a one-GPU box."""
    from schnorr_amd import _lib
    n = 9000 + 13
    d = O.keygen_sign_single(n, 17, nthreads=8)
    H.tamper(d, period=9)
    want = O.verify_single(d["u"], d["R"], d["PK"], d["m"], nthreads=8)
    assert engine.initialized_devices() == [0]
    assert np.array_equal(engine.verify_single_multi(d["u"], d["R"], d["PK"], d["m"]), want)
    dd = O.keygen_sign_double(4100, 18, nthreads=8)
    H.tamper(dd, period=6)
    want_d = O.verify_double(dd["u"], dd["R"], dd["Rp"], dd["PK"], dd["PKp"], dd["m"], nthreads=8)
    dv = O.keygen_sign_vargen(4100, 19, nthreads=8)
    H.tamper(dv, period=6)
    want_v = O.verify_vargen(dv["u"], dv["R"], dv["PK"], dv["Gen"], dv["m"], nthreads=8)
    os.environ["DSV_MULTI_SHARDS"] = "4"
    try:
        assert np.array_equal(engine.verify_single_multi(d["u"], d["R"], d["PK"], d["m"]), want)
        assert np.array_equal(engine.verify_double_multi(dd["u"], dd["R"], dd["Rp"], dd["PK"], dd["PKp"],
                                                         dd["m"]), want_d)
        assert np.array_equal(engine.verify_vargen_multi(dv["u"], dv["R"], dv["PK"], dv["Gen"], dv["m"]),
                              want_v)
    finally:
        del os.environ["DSV_MULTI_SHARDS"]
    # a second device that does not exist is refused, the first stays usable
    with pytest.raises(_lib.DsvError):
        engine.init(engine._lib.load().dsv_device_count())
    with pytest.raises(_lib.DsvError):
        engine.set_device(1 if engine._lib.load().dsv_device_count() < 2 else 15)
    engine.set_device(0)
    assert np.array_equal(engine.verify_single(d["u"][:100], d["R"][:100], d["PK"][:100], d["m"][:100]),
                          want[:100])


def test_many_device_callers_then_shutdown_contract(engine):
    """include/dsv.h "Devices and threads": four host threads enqueue large (sub-batched) calls on
    their own streams at once — each gets its own pair of internal streams — while a fifth uses
    the host entry point; afterwards dsv_shutdown_device refuses new calls until dsv_init."""
    import torch
    from schnorr_amd import _lib
    from schnorr_amd import workload as W
    n = (1 << 17) + 321
    batches = [W.gen_single(n, seed=300 + t) for t in range(4)]
    hb = W.gen_single(30000, seed=9)
    host = {k: hb[k].cpu().numpy() for k in ("u", "R", "PK", "m")}
    host_want = hb["expected"].cpu().numpy()
    torch.cuda.synchronize()
    errors = []

    def dev_worker(t):
        try:
            b = batches[t]
            st = torch.cuda.Stream()
            ws = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
            ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
            torch.cuda.synchronize()
            for _ in range(4):
                ok.zero_()
                st.wait_stream(torch.cuda.current_stream())
                engine.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws, stream=st)
                st.synchronize()
                if not torch.equal(ok, b["expected"]):
                    errors.append("device thread %d: wrong verdicts" % t)
        except Exception as e:  # noqa: BLE001
            errors.append("device thread %d: %r" % (t, e))

    def host_worker():
        try:
            for _ in range(4):
                if not np.array_equal(engine.verify_single(host["u"], host["R"], host["PK"], host["m"]),
                                      host_want):
                    errors.append("host thread: wrong verdicts")
        except Exception as e:  # noqa: BLE001
            errors.append("host thread: %r" % (e,))

    threads = [threading.Thread(target=dev_worker, args=(t,)) for t in range(4)]
    threads.append(threading.Thread(target=host_worker))
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    torch.cuda.synchronize()
    engine.shutdown(0)
    assert engine.initialized_devices() == []
    b = batches[0]
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(_lib.DsvError):
        engine.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    with pytest.raises(_lib.DsvError):
        engine.verify_single(host["u"], host["R"], host["PK"], host["m"])
    engine.init(0)
    engine.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    torch.cuda.synchronize()
    assert torch.equal(ok, b["expected"])


def test_more_caller_streams_than_lanes(engine):
    """A context owns 8 sub-batch lanes; the 9th distinct caller stream shares one by hashing.
    Twelve streams with a large (sub-batched) call each, all in flight together: every call still
    behaves as one enqueue on its own stream."""
    import torch
    from schnorr_amd import workload as W
    n = (1 << 17) + 77
    b = W.gen_single(n, seed=41)
    streams = [torch.cuda.Stream() for _ in range(12)]
    oks = [torch.zeros(n, dtype=torch.uint8, device="cuda:0") for _ in streams]
    wss = [torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device="cuda:0") for _ in streams]
    torch.cuda.synchronize()
    for st, ok, ws in zip(streams, oks, wss):
        engine.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws, stream=st)
    for st in streams:
        st.synchronize()
    for ok in oks:
        assert torch.equal(ok, b["expected"])


def test_bench_spawns_its_own_ranks_and_runs_the_mixed_config():
    """`python bench.py --gpus 2` from a plain shell (no torchrun, WORLD_SIZE
    unset) starts two fresh ranks itself.  On this one-GPU box both ranks share GPU 0 and the
    gather runs over gloo (RCCL refuses two ranks on one device); the code path is the one the
    8-GPU run takes: per-rank shards of one StdRng stream, HIP engine, all_gather inside the
    timed region, and the configs[4] mixed batch with its device-side split."""
    drop = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
    clean = {k: v for k, v in os.environ.items() if k not in drop}
    clean.update({"DSV_BENCH_DEVICE": "0", "DSV_BENCH_BACKEND": "gloo"})

    def run(args):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=clean,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-1500:]
        return json.loads(lines[0])

    d = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "15"])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["launcher"] == "self-spawned"
    assert d["backend"] == "gloo" and d["value"] > 0 and d["scaling"] == "weak"
    assert d["mixed"]["n_gpus"] == 2 and d["mixed"]["value"] > 0
    m = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "15", "--config", "mixed"])
    assert m["world_size"] == 2 and m["value"] > 0 and "configs[4]" in m["config"]["workload"]


def test_bench_collectives_over_rccl_with_one_rank():
    """The RCCL half of the multi-GPU path on a one-GPU box: DSV_BENCH_FORCE_DIST=1 creates the
    "nccl" process group for a single rank, so the timed steps run the real RCCL all_gather /
    all_reduce / barrier (with N ranks only the communicator is wider)."""
    drop = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DSV_BENCH_BACKEND")
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env["DSV_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--log2-batch", "16", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][0])
    assert d["backend"] == "rccl" and d["rccl_version"] and d["world_size"] == 1
    assert d["config"]["collective"].startswith("all_gather") and d["mixed"]["value"] > 0


def _sharded_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from schnorr_amd import distributed as D
    from schnorr_amd import engine as E
    E.init(0)
    n = 777
    d = O.keygen_sign_single(n, 2321)
    H.tamper(d, period=5)
    full = D.verify_single_sharded(d["u"], d["R"], d["PK"], d["m"], verify_fn=E.verify_single,
                                   to_tensor=lambda a: torch.from_numpy(a))
    want = O.verify_single(d["u"], d["R"], d["PK"], d["m"])
    ds = O.keygen_sign_single(300, 5)
    dd = O.keygen_sign_double(211, 6)
    H.tamper(ds, period=4)
    H.tamper(dd, period=3)
    kinds = np.array([0] * 300 + [1] * 211)
    np.random.default_rng(1).shuffle(kinds)
    mixed = D.verify_mixed_sharded(kinds, (ds["u"], ds["R"], ds["PK"], ds["m"]),
                                   (dd["u"], dd["R"], dd["Rp"], dd["PK"], dd["PKp"], dd["m"]),
                                   E.verify_single, E.verify_double, lambda a: torch.from_numpy(a))
    wm = np.zeros(511, np.uint8)
    wm[kinds == 0] = O.verify_single(ds["u"], ds["R"], ds["PK"], ds["m"])
    wm[kinds == 1] = O.verify_double(dd["u"], dd["R"], dd["Rp"], dd["PK"], dd["PKp"], dd["m"])
    q.put((rank, bool(np.array_equal(full.numpy(), want)), bool(np.array_equal(mixed.numpy(), wm))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_entry_points_with_the_hip_engine_in_two_ranks():
    """schnorr_amd/distributed.py's cooperative entry points with the HIP engine as verify_fn in two
    spawned ranks sharing GPU 0 (gloo gather of host verdicts)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]


def test_challenge_hash_ragged_waves_and_extreme_inputs(engine):
    """The hashes of a wave cooperate through the matrix cores (hades_mfma.h): batch sizes that
    leave a wave / a workgroup partly empty, and field elements at the ends of the range, must give
    the oracle's challenges bit for bit (single and double hash)."""
    rng = np.random.default_rng(99)

    def felts(n):
        vals = [int.from_bytes(rng.bytes(40), "little") % M.Q for _ in range(n)]
        for k, v in enumerate([0, 1, M.Q - 1, M.Q - 2, (1 << 254) - 1, 1 << 254, (1 << 128) - 1]):
            if k < n:
                vals[(5 * k) % n] = v
        return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(n, 32)

    for n in (1, 2, 31, 33, 63, 64, 65, 127, 255, 256, 257, 300):
        R = np.concatenate([felts(n), felts(n)], axis=1)
        Rp = np.concatenate([felts(n), felts(n)], axis=1)
        m = felts(n)
        assert np.array_equal(engine.challenge_single(R, m), O.challenge_single(R, m)), n
        assert np.array_equal(engine.challenge_double(R, Rp, m), O.challenge_double(R, Rp, m)), n


def test_fixed_base_windows_at_their_extreme_digits(engine):
    """Scalars whose signed fixed-base windows (dsv_fixed_window_bits() wide) all sit at an extreme
    digit — +-2^(bits-1), +-(2^(bits-1) - 1), 0, 1 — through key derivation for both generators,
    against the oracle's generic scalar multiplication."""
    bits = engine.fixed_window_bits()
    half = 1 << (bits - 1)
    windows = (253 + bits - 1) // bits
    scalars = []
    for pat in (half, half - 1, half + 1, (1 << bits) - 1, 1, 0):
        v = sum(pat << (bits * w) for w in range(windows)) % M.R_ORDER
        scalars += [v, (M.R_ORDER - v) % M.R_ORDER]
    rng = np.random.default_rng(3)
    for _ in range(8):      # random windows drawn from the extreme set only
        v = sum(int(rng.choice([0, 1, half - 1, half, half + 1, (1 << bits) - 1])) << (bits * w) for w in range(windows))
        scalars.append(v % M.R_ORDER)
    SK = np.stack([np.frombuffer(M.le32(x), np.uint8) for x in scalars])
    for which, gen in ((0, M.GEN), (1, M.GEN_NUMS)):
        G = np.frombuffer(M.le32(gen[0]) + M.le32(gen[1]), np.uint8)
        want = O.scalar_mul(SK, np.broadcast_to(G, (len(scalars), 64)).copy())
        got = engine.public_keys(SK, which)
        assert np.array_equal(got, want), which
