"""GPU tests of round 6: failure localisation of the batch fast accept ("bisect on
failure" done as sub-groups that share one set of launches: schnorr_amd/csrc/rlc.h, dsv_rlc.hip) and its
enqueue-only control (per-signature kernels gated by the aggregate's flag words, the verdict written by
a kernel).  The contract is unchanged: the verdict vector of `PublicKey::verify`
(/root/reference/src/keys/public.rs:121-130; :222-244, :401-415 for the other schemes) item by item —
the ORACLE's — whatever the sub-group count, history or sample says."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import harness as H
import oracle_lib as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = {"single": ("u", "R", "PK", "m"), "double": ("u", "R", "Rp", "PK", "PKp", "m"),
        "vargen": ("u", "R", "PK", "Gen", "m")}


def _signed(n, seed, scheme="single"):
    d = getattr(O, "keygen_sign_" + scheme)(n, seed, nthreads=8)
    return {k: d[k] for k in COLS[scheme]}


def _oracle(a, scheme):
    return getattr(O, "verify_" + scheme)(*[a[k] for k in COLS[scheme]], nthreads=8)


def _run(engine, a, scheme, window_bits=0, accepted_out=None):
    n = len(a["u"])
    t = [torch.from_numpy(np.ascontiguousarray(a[k])).to(DEV) for k in COLS[scheme]]
    ok = torch.full((n,), 7, dtype=torch.uint8, device=DEV)
    ws = torch.empty(engine.rlc_workspace_bytes(n, window_bits), dtype=torch.uint8, device=DEV)
    acc = getattr(engine, "verify_%s_rlc_dev" % scheme)(*t, ok, ws, window_bits=window_bits, accepted_out=accepted_out)
    torch.cuda.synchronize()
    return acc, ok.cpu().numpy()


@pytest.fixture
def forced_groups(engine):
    """sub-groups forced for the test, the automatic choice restored afterwards"""
    def force(g):
        engine.rlc_subgroups(g)
    yield force
    engine.rlc_subgroups(0)


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
@pytest.mark.parametrize("groups", [2, 4, 7])
def test_sub_groups_give_the_oracles_verdicts(engine, forced_groups, scheme, groups):
    """a wrong item in each sub-group in turn, two in different sub-groups, none; ragged last sub-group"""
    n = 3000 + 37
    forced_groups(groups)
    d = _signed(n, 600 + groups, scheme)
    acc, ok = _run(engine, d, scheme, 8)
    assert acc and ok.all()
    sub = engine.rlc_plan_info(scheme, n, 8, groups)["sub"]
    G = engine.rlc_plan_info(scheme, n, 8, groups)["groups"]
    assert (G - 1) * sub < n <= G * sub and n % sub != 0
    victims = [g * sub + (17 * g) % min(sub, n - g * sub) for g in range(G)] + [n - 1]
    for v in victims:
        a = {k: x.copy() for k, x in d.items()}
        a["u"][v, 5] ^= 0x02
        want = _oracle(a, scheme)
        assert want.sum() == n - 1 and not want[v]
        acc, ok = _run(engine, a, scheme, 8)
        assert not acc and np.array_equal(ok, want), v
    a = {k: x.copy() for k, x in d.items()}
    a["m"][victims[0], 1] ^= 0x40
    a["PK"][victims[-1]] = d["PK"][victims[-1] - 1]
    want = _oracle(a, scheme)
    assert want.sum() == n - 2
    acc, ok = _run(engine, a, scheme, 8)
    assert not acc and np.array_equal(ok, want)
    # the harness's tamper classes, all over the batch
    a = {k: x.copy() for k, x in d.items()}
    H.tamper(a, period=11)
    want = _oracle(a, scheme)
    acc, ok = _run(engine, a, scheme, 12)
    assert not acc and np.array_equal(ok, want)


def test_a_small_order_component_fails_only_its_own_sub_group(engine, forced_groups):
    """a key with an order-8 component whose item is VALID (the torsion parts cancel): its sub-group must not be
    decided by its aggregate; the verdicts are the oracle's; seen per sub-group through DSV_RLC_TRACE"""
    code = r"""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O, test_halfgcd as TH, test_gpu_rlc as T
from schnorr_amd import engine as E
E.init(0)
n = 4000
d = O.keygen_sign_single(n, 611, nthreads=8)
a = {k: d[k].copy() for k in ("u", "R", "PK", "m")}
rows = T._torsion_rows(TH.order8_point(), TH.rnd, 1, True)
for k in rows:
    a[k][2500] = rows[k][0]
want = O.verify_single(a["u"], a["R"], a["PK"], a["m"], nthreads=8)
assert want.all()
E.rlc_subgroups(4)
t = [torch.from_numpy(np.ascontiguousarray(a[k])).to("cuda:0") for k in ("u", "R", "PK", "m")]
ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
ws = torch.empty(E.rlc_workspace_bytes(n, 8), dtype=torch.uint8, device="cuda:0")
acc = E.verify_single_rlc_dev(*t, ok, ws, window_bits=8)
assert not acc and np.array_equal(ok.cpu().numpy(), want)
print("done")
"""
    env = dict(os.environ, DSV_RLC_TRACE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    lines = [x for x in r.stderr.splitlines() if "[dsv rlc]" in x]
    assert len(lines) == 4, lines
    assert [("accepted" in x) for x in lines] == [True, True, False, True], lines
    assert "subgroup-test" in lines[2]


def test_history_follows_the_calls(engine):
    """the device keeps the counter itself: 8 after a call with a rejected aggregate, one less after a call whose
    aggregates all accepted; groups too small for an aggregate leave it alone; while it is > 0 a group of 2^18
    items runs in two sub-groups, at 0 in one"""
    n = 1 << 18
    d = _signed(4096, 620)
    big = {k: np.tile(v, (n // 4096, 1)) for k, v in d.items()}
    bad = {k: v.copy() for k, v in big.items()}
    bad["u"][n // 2 + 5, 0] ^= 1
    engine.rlc_history(0, 0)
    acc, ok = _run(engine, big, "single")
    assert acc and ok.all() and engine.rlc_history(0) == 0
    acc, ok = _run(engine, bad, "single")
    assert not acc and ok.sum() == n - 1 and not ok[n // 2 + 5]
    assert engine.rlc_history(0) == 8
    for left in (7, 6):
        acc, ok = _run(engine, big, "single")
        assert acc and ok.all() and engine.rlc_history(0) == left
    acc, ok = _run(engine, d, "single")            # 4096 items: no aggregate, no change
    assert not acc and ok.all() and engine.rlc_history(0) == 6
    acc, ok = _run(engine, bad, "single")          # rejected again, in sub-groups this time
    assert not acc and ok.sum() == n - 1 and engine.rlc_history(0) == 8
    engine.rlc_history(0, 1)


def test_accepted_through_memory_the_device_writes(engine):
    """`accepted` in pinned host memory or in device memory: written by a kernel, the call returns None at once"""
    n = 2500
    d = _signed(n, 630)
    bad = {k: v.copy() for k, v in d.items()}
    bad["m"][99, 0] ^= 4
    for make in (lambda: torch.full((1,), 5, dtype=torch.int32).pin_memory(),
                 lambda: torch.full((1,), 5, dtype=torch.int32, device=DEV)):
        word = make()
        acc, ok = _run(engine, d, "single", 8, accepted_out=word)
        assert acc is None and int(word.cpu()[0]) == 1 and ok.all()
        acc, ok = _run(engine, bad, "single", 8, accepted_out=word)
        assert acc is None and int(word.cpu()[0]) == 0 and ok.sum() == n - 1
    with pytest.raises(ValueError):
        _run(engine, d, "single", 8, accepted_out=torch.zeros(1, dtype=torch.int32))   # pageable: not through this argument


def test_one_wrong_signature_in_a_full_size_batch(engine):
    """2^20 signatures with ONE wrong (BASELINE configs[1]'s size): verdicts = the construction pattern, whether
    the call runs as one group (history 0) or in sub-groups; an oracle sample around the wrong item"""
    from schnorr_amd import workload as W
    n = 1 << 20
    b = W.gen_single(n, seed=77, tamper=False)
    victim = 700_001
    b["u"][victim, 2] ^= 0x20
    b["expected"][victim] = 0
    ws = torch.empty(engine.rlc_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    ok = torch.zeros(n, dtype=torch.uint8, device=DEV)
    for history in (0, 8):
        engine.rlc_history(0, history)
        ok.zero_()
        assert not engine.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
        assert torch.equal(ok, b["expected"])
    lo = victim - 100
    cut = {k: b[k][lo:lo + 256].cpu().numpy() for k in COLS["single"]}
    assert np.array_equal(_oracle(cut, "single"), ok[lo:lo + 256].cpu().numpy())
    engine.rlc_history(0, 1)


def test_mixed_batch_beyond_one_group_per_kind(engine):
    """a mixed batch just above 2^22 items — its kinds' counts are single groups of ~2^21
    items while the workspace is sized for n: all valid -> accepted, one wrong double -> the pattern"""
    from schnorr_amd import workload as W
    n = (1 << 22) + (1 << 12)
    b = W.gen_mixed(n, seed=31, tamper=False)
    ws = torch.empty(engine.mixed_rlc_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    ok = torch.zeros(n, dtype=torch.uint8, device=DEV)
    args = [b[k] for k in ("kinds", "u", "R", "Rp", "PK", "PKp", "m")] + [b["n_double"]]
    engine.rlc_history(0, 0)
    assert engine.verify_mixed_rlc_dev(*args, ok, ws)
    assert bool(ok.all())
    victim = 3_000_001   # odd: a double item
    b["PKp"][victim] = b["PKp"][victim - 2]
    ok.zero_()
    assert not engine.verify_mixed_rlc_dev(*args, ok, ws)
    assert int(ok.sum()) == n - 1 and int(ok[victim]) == 0
    engine.rlc_history(0, 1)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 8192, 8193])
def test_var_generator_sixteen_lane_kernel(engine, n):
    """`PublicKeyVarGen::verify` (/root/reference/src/keys/public.rs:401-415) through k_verify_var_hex (launches of
    at most 2^13 items: sixteen lanes per signature) and, one item beyond, through the one-lane kernel: the
    oracle's verdicts on the harness's tamper classes, on malformed items and on the identity as key"""
    base = 700
    d = _signed(min(n, base), 640 + n % 97, "vargen")
    if n > base:
        reps = -(-n // base)
        d = {k: np.tile(v, (reps, 1))[:n].copy() for k, v in d.items()}
    if n >= 16:
        H.tamper(d, period=5)
        d["u"][3] = 0xFF                         # u >= r
        d["PK"][7, 31] |= 0x80                   # a coordinate >= q
        d["Gen"][9, :32] = 0
        d["Gen"][9, 32:] = 0
        d["Gen"][9, 32] = 1                      # the identity as generator
    want = _oracle(d, "vargen")
    t = [torch.from_numpy(np.ascontiguousarray(d[k])).to(DEV) for k in COLS["vargen"]]
    ok = torch.full((n,), 7, dtype=torch.uint8, device=DEV)
    ws = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device=DEV)
    engine.verify_vargen_dev(*t, ok, ws)
    torch.cuda.synchronize()
    assert np.array_equal(ok.cpu().numpy(), want)
    if n >= 16:
        assert 0 < want.sum() < n


@pytest.mark.parametrize("scheme", ["double", "vargen"])
def test_double_and_var_generator_aggregates_start_at_2_14_items(engine, scheme):
    """automatic window bits: a device-resident double / var-generator batch is decided by its aggregate from
    2^14 items on (rlc.h: rlc_min_auto; a single-signature batch from 2^17), one item fewer takes the
    per-signature kernels; verdicts are the oracle's either way"""
    base = _signed(1024, 650, scheme)
    for n, expect in (((1 << 14) - 1, False), (1 << 14, True), ((1 << 14) + 333, True)):
        d = {k: np.tile(v, (-(-n // 1024), 1))[:n].copy() for k, v in base.items()}
        engine.rlc_history(0, 0)
        acc, ok = _run(engine, d, scheme)
        assert acc == expect and ok.all(), n
        d["m"][n // 3, 4] ^= 1
        want = np.ones(n, np.uint8)
        want[n // 3] = 0
        acc, ok = _run(engine, d, scheme)
        assert not acc and np.array_equal(ok, want), n
        cut = slice(n // 3 - 20, n // 3 + 20)
        assert np.array_equal(_oracle({k: v[cut] for k, v in d.items()}, scheme), want[cut])
    engine.rlc_history(0, 1)


def test_two_groups_in_sub_groups(engine):
    """2^22 + 2^18 + 5 signatures while the history says "batches fail": two groups, each cut into sub-groups (the
    last one ragged), one wrong signature in each group — the construction pattern; then all valid -> accepted"""
    from schnorr_amd import workload as W
    n = (1 << 22) + (1 << 18) + 5
    b = W.gen_single(n, seed=9, tamper=False)
    ws = torch.empty(engine.rlc_workspace_bytes(n), dtype=torch.uint8, device=DEV)
    ok = torch.zeros(n, dtype=torch.uint8, device=DEV)
    engine.rlc_history(0, 8)
    assert engine.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    assert bool(ok.all())
    want = torch.ones(n, dtype=torch.uint8, device=DEV)
    for victim in (1_234_567, n - 2):
        b["m"][victim, 7] ^= 0x08
        want[victim] = 0
    ok.zero_()
    engine.rlc_history(0, 8)
    assert not engine.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    assert torch.equal(ok, want)
    # the same two groups guarded: each runs its single aggregate, then (rejected) its second stage
    ok.zero_()
    engine.rlc_history(0, 0)
    engine.rlc_history_long(0, 100)
    assert not engine.verify_single_rlc_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
    assert torch.equal(ok, want)
    assert engine.rlc_history(0) == 8 and engine.rlc_history_long(0) == 128
    engine.rlc_history(0, 1)
    engine.rlc_history_long(0, 0)


def test_device_calls_from_several_threads_and_streams(engine):
    """four host threads, each with its own stream, workspace and verdict buffer, enqueue fast-accept calls at the
    same time (valid and invalid batches alternating): every call's verdicts and `accepted` are its own"""
    import threading
    n = 6000
    d = _signed(n, 660)
    bad = {k: v.copy() for k, v in d.items()}
    bad["u"][4321, 9] ^= 0x80
    want_bad = _oracle(bad, "single")
    assert want_bad.sum() == n - 1
    dev = {name: [torch.from_numpy(np.ascontiguousarray(x[k])).to(DEV) for k in COLS["single"]]
           for name, x in (("good", d), ("bad", bad))}
    errors = []

    def worker(t):
        try:
            stream = torch.cuda.Stream(device=DEV)
            ok = torch.zeros(n, dtype=torch.uint8, device=DEV)
            ws = torch.empty(engine.rlc_workspace_bytes(n, 8), dtype=torch.uint8, device=DEV)
            word = torch.zeros(1, dtype=torch.int32).pin_memory()
            for call in range(6):
                which = "bad" if (call + t) & 1 else "good"
                with torch.cuda.stream(stream):
                    ok.fill_(9)
                    engine.verify_single_rlc_dev(*dev[which], ok, ws, stream=stream, window_bits=8, accepted_out=word)
                stream.synchronize()
                got = ok.cpu().numpy()
                exp = want_bad if which == "bad" else np.ones(n, np.uint8)
                if int(word[0]) != (which == "good") or not np.array_equal(got, exp):
                    errors.append("thread %d call %d (%s): accepted=%d, %d verdicts differ" % (
                        t, call, which, int(word[0]), int((got != exp).sum())))
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    engine.rlc_history(0, 1)


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_guarded_groups(engine, scheme):
    """history 0, long history > 0 ("batches fail now and then"): one aggregate plus a second stage of sub-group
    aggregates that runs only where the first rejected.  Verdicts = the oracle's; `accepted` and both counters
    follow the decisive stage."""
    n = 5000 + 13
    d = _signed(n, 670, scheme)
    bad = {k: v.copy() for k, v in d.items()}
    bad["u"][n - 7, 11] ^= 0x04
    want = _oracle(bad, scheme)
    assert want.sum() == n - 1
    engine.rlc_history(0, 0)
    engine.rlc_history_long(0, 20)
    engine.rlc_subgroups(5)       # (the second stage's sub-groups; the first stage is one aggregate)
    acc, ok = _run(engine, d, scheme, 8)
    assert acc and ok.all()
    assert engine.rlc_history(0) == 0 and engine.rlc_history_long(0) == 19
    acc, ok = _run(engine, bad, scheme, 8)
    assert not acc and np.array_equal(ok, want)
    assert engine.rlc_history(0) == 8 and engine.rlc_history_long(0) == 128
    # tampered throughout, guarded again
    engine.rlc_history(0, 0)
    a = {k: v.copy() for k, v in d.items()}
    H.tamper(a, period=13)
    want = _oracle(a, scheme)
    acc, ok = _run(engine, a, scheme, 12)
    assert not acc and np.array_equal(ok, want)
    engine.rlc_subgroups(0)
    engine.rlc_history(0, 1)
    engine.rlc_history_long(0, 0)


def test_guarded_second_stage_runs_only_after_a_reject():
    """seen through DSV_RLC_TRACE: a valid batch -> first stage accepted, second stage not needed; one wrong item ->
    first stage rejected ("sum"), the second stage's sub-groups all accepted but the one that holds it"""
    code = r"""
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_lib as O
from schnorr_amd import engine as E
E.init(0)
n = 4000
d = O.keygen_sign_single(n, 671, nthreads=8)
cols = ("u", "R", "PK", "m")
ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
ws = torch.empty(E.rlc_workspace_bytes(n, 8), dtype=torch.uint8, device="cuda:0")
E.rlc_subgroups(4)
for label, victim in (("valid", None), ("one", 3100)):
    a = {k: d[k].copy() for k in cols}
    if victim is not None:
        a["m"][victim, 0] ^= 1
    E.rlc_history(0, 0); E.rlc_history_long(0, 32)
    sys.stderr.write("CALL %s\n" % label); sys.stderr.flush()
    t = [torch.from_numpy(np.ascontiguousarray(a[k])).to("cuda:0") for k in cols]
    acc = E.verify_single_rlc_dev(*t, ok, ws, window_bits=8)
    want = np.ones(n, np.uint8)
    if victim is not None:
        want[victim] = 0
    assert acc == (victim is None) and np.array_equal(ok.cpu().numpy(), want)
print("done")
"""
    env = dict(os.environ, DSV_RLC_TRACE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    calls, cur = {}, None
    for line in r.stderr.splitlines():
        if line.startswith("CALL "):
            cur = line[5:]
            calls[cur] = []
        elif "[dsv rlc]" in line and cur:
            calls[cur].append(line)
    assert any("first stage" in x and "accepted" in x for x in calls["valid"]) and any("not needed" in x for x in calls["valid"])
    assert any("first stage" in x and " sum" in x for x in calls["one"])
    second = [x for x in calls["one"] if "second stage" in x]
    assert len(second) == 4 and sum("accepted" in x for x in second) == 3 and sum(" sum" in x for x in second) == 1, second


def test_tiny_and_empty_batches_through_the_fast_accept(engine, forced_groups):
    """n = 0 (nothing enqueued, `accepted` cleared wherever it lives), n = 1 .. 5 with more sub-groups asked for than
    there are items, explicit window bits: the oracle's verdicts"""
    d = _signed(5, 680)
    t_all = {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(DEV) for k in COLS["single"]}
    ws = torch.empty(engine.rlc_workspace_bytes(8, 8), dtype=torch.uint8, device=DEV)
    ok = torch.zeros(8, dtype=torch.uint8, device=DEV)
    for word in (torch.full((1,), 7, dtype=torch.int32).pin_memory(), torch.full((1,), 7, dtype=torch.int32, device=DEV)):
        empty = [t_all[k][:0] for k in COLS["single"]]
        assert engine.verify_single_rlc_dev(*empty, ok[:0], ws, window_bits=8, accepted_out=word) is None
        torch.cuda.synchronize()
        assert int(word.cpu()[0]) == 0
    assert engine.verify_single_rlc_dev(*[t_all[k][:0] for k in COLS["single"]], ok[:0], ws, window_bits=8) is False
    forced_groups(16)
    for n in (1, 2, 3, 5):
        cut = {k: d[k][:n].copy() for k in COLS["single"]}
        acc, got = _run(engine, cut, "single", 8)
        assert acc and got.all(), n
        cut["u"][n - 1, 0] ^= 1
        acc, got = _run(engine, cut, "single", 8)
        assert not acc and np.array_equal(got, _oracle(cut, "single")), n
