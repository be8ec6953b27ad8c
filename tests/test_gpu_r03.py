"""GPU tests of what round 3 added: projective (u, v, z) entry points for every scheme (host, device
pointers, multi-device), wire records resident in HBM, index vectors that are only dereferenced
where the split wrote them, the half-gcd's exact comparison, world sizes 4 and 8.  Parity is always
against the CPU oracle (oracle/)."""
import json
import os
import subprocess
import sys

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


def _ext_case(scheme, n, seed):
    """(affine batch, uvz arrays in entry-point order, oracle verdicts on the 160-byte ext points)"""
    rng = np.random.default_rng(seed)
    if scheme == "single":
        d = O.keygen_sign_single(n, seed, nthreads=8)
        names = ("R", "PK")
    elif scheme == "double":
        d = O.keygen_sign_double(n, seed, nthreads=8)
        names = ("R", "Rp", "PK", "PKp")
    else:
        d = O.keygen_sign_vargen(n, seed, nthreads=8)
        names = ("R", "PK", "Gen")
    H.tamper(d, period=7)
    uvz, ext = {}, {}
    for j, k in enumerate(names):
        # a z = 0 and a non-canonical z per point array, on items that would otherwise verify
        uvz[k], ext[k] = H.projective(d[k], rng, zero_z={3 + 14 * j}, noncanon_z={10 + 14 * j})
    if scheme == "single":
        want = O.verify_single_ext(d["u"], ext["R"], ext["PK"], d["m"])
    elif scheme == "double":
        want = O.verify_double_ext(d["u"], ext["R"], ext["Rp"], ext["PK"], ext["PKp"], d["m"])
    else:
        want = O.verify_vargen_ext(d["u"], ext["R"], ext["PK"], ext["Gen"], d["m"])
    for j in range(len(names)):
        want[3 + 14 * j] = 0            # z = 0: the reference would panic in to_hash_inputs; here 0
        assert want[10 + 14 * j] == 0   # the oracle already rejects the non-canonical z
    return d, [uvz[k] for k in names], want


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_projective_entry_points_match_the_oracle(engine, scheme):
    """dsv_verify_*_ext / _ext_multi / _ext_dev: to_hash_inputs on the device (one inversion per
    signature at most), random z per point, tampered items, z = 0 and non-canonical z items."""
    import torch
    n = 300 + 11
    d, pts, want = _ext_case(scheme, n, {"single": 5, "double": 6, "vargen": 7}[scheme])
    assert 0 < want.sum() < n
    host = getattr(engine, "verify_%s_ext" % scheme)
    got = host(d["u"], *pts, d["m"])
    assert np.array_equal(got, want)
    os.environ["DSV_MULTI_SHARDS"] = "3"          # the sharding arithmetic (x96 offsets) on one GPU
    try:
        # below nd * 1024 items the multi entry point takes one device: use a tiled batch
        reps = 12
        tile = lambda a: np.tile(a, (reps, 1))
        got_m = host(tile(d["u"]), *[tile(p) for p in pts], tile(d["m"]), multi=True)
    finally:
        del os.environ["DSV_MULTI_SHARDS"]
    assert np.array_equal(got_m, np.tile(want, reps))
    ok = torch.full((n,), 9, dtype=torch.uint8, device="cuda:0")
    ws = torch.full((engine.ext_workspace_bytes(n),), 0xFF, dtype=torch.uint8, device="cuda:0")
    getattr(engine, "verify_%s_ext_dev" % scheme)(_dev(d["u"]), *[_dev(p) for p in pts], _dev(d["m"]), ok, ws)
    torch.cuda.synchronize()
    assert np.array_equal(ok.cpu().numpy(), want)


def test_projective_large_batch_shares_one_inversion_among_eight_items(engine):
    """n >= 2^15: a lane normalises eight items with ONE inversion (Montgomery's trick across items);
    an offending z must not poison its lane mates.  Checked against the affine entry point on the
    points the oracle normalised, plus an oracle sample."""
    import torch
    n = (1 << 15) + 37
    base = 257
    d = O.keygen_sign_single(base, 41, nthreads=8)
    H.tamper(d, period=5)
    rng = np.random.default_rng(8)
    R_uvz, R_ext = H.projective(d["R"], rng, zero_z={0, 100}, noncanon_z={1})
    PK_uvz, PK_ext = H.projective(d["PK"], rng, zero_z={200})
    want = O.verify_single_ext(d["u"], R_ext, PK_ext, d["m"])
    want[[0, 100, 200]] = 0
    reps = -(-n // base)
    tile = lambda a: np.tile(a, (reps, 1))[:n]
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(engine.ext_workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    engine.verify_single_ext_dev(_dev(tile(d["u"])), _dev(tile(R_uvz)), _dev(tile(PK_uvz)), _dev(tile(d["m"])), ok, ws)
    torch.cuda.synchronize()
    assert np.array_equal(ok.cpu().numpy(), np.tile(want, reps)[:n])
    assert 0 < want.sum() < base
    # and through the host pipeline (chunked; the ext scratch lives in the slot's extra area)
    got = engine.verify_single_ext(tile(d["u"]), tile(R_uvz), tile(PK_uvz), tile(d["m"]))
    assert np.array_equal(got, np.tile(want, reps)[:n])


@pytest.mark.parametrize("scheme", ["single", "double", "vargen"])
def test_wire_records_resident_in_device_memory(engine, scheme):
    """dsv_verify_*_wire_dev against the oracle's from_bytes + verify, incl. undecodable records."""
    import torch
    n = 500 + 3
    seed = {"single": 15, "double": 16, "vargen": 17}[scheme]
    if scheme == "single":
        d = O.keygen_sign_single(n, seed, nthreads=8)
        H.tamper(d, period=9)
        sig = np.concatenate([d["u"], engine.compress_points(d["R"])], axis=1)
        pk = engine.compress_points(d["PK"])
    elif scheme == "double":
        d = O.keygen_sign_double(n, seed, nthreads=8)
        H.tamper(d, period=9)
        sig = np.concatenate([d["u"], engine.compress_points(d["R"]), engine.compress_points(d["Rp"])], axis=1)
        pk = np.concatenate([engine.compress_points(d["PK"]), engine.compress_points(d["PKp"])], axis=1)
    else:
        d = O.keygen_sign_vargen(n, seed, nthreads=8)
        H.tamper(d, period=9)
        sig = np.concatenate([d["u"], engine.compress_points(d["R"])], axis=1)
        pk = np.concatenate([engine.compress_points(d["PK"]), engine.compress_points(d["Gen"])], axis=1)
    sig, pk = np.ascontiguousarray(sig), np.ascontiguousarray(pk)
    sig[5, 40] ^= 1          # almost surely no longer a decodable point
    pk[7, -1] |= 0x7f        # v >= q
    want = getattr(O, "verify_%s_wire" % scheme)(sig, pk, d["m"])
    assert 0 < want.sum() < n and want[7] == 0
    ok = torch.full((n,), 5, dtype=torch.uint8, device="cuda:0")
    ws = torch.full((engine.wire_workspace_bytes(n),), 0xFF, dtype=torch.uint8, device="cuda:0")
    getattr(engine, "verify_%s_wire_dev" % scheme)(_dev(sig), _dev(pk), _dev(d["m"]), ok, ws)
    torch.cuda.synchronize()
    assert np.array_equal(ok.cpu().numpy(), want)
    assert np.array_equal(getattr(engine, "verify_%s_wire" % scheme)(sig, pk, d["m"]), want)


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


def test_mixed_batch_out_of_contract_calls_on_a_poisoned_workspace(engine):
    """with a wrong n_double or an unknown kind the tail of an index vector is
    never written by the split.  Every call here gets a FRESH workspace filled with 0xFF (index
    0xFFFFFFFF, row offset ~2^37): dereferencing one unwritten entry would fault.  Verdicts: all 0
    for a count mismatch; the in-contract call on the same poisoned workspace is exact."""
    import torch
    n = 3000 + 5
    kinds, cols, want, nd = _mixed_arrays(n, 23)
    t = {k: _dev(v) for k, v in cols.items()}

    def run(kv, ndecl):
        ok = torch.full((n,), 7, dtype=torch.uint8, device="cuda:0")
        ws = torch.full((engine.mixed_workspace_bytes(n),), 0xFF, dtype=torch.uint8, device="cuda:0")
        engine.verify_mixed_dev(_dev(kv), t["u"], t["R"], t["Rp"], t["PK"], t["PKp"], t["m"], ndecl, ok, ws)
        torch.cuda.synchronize()
        return ok.cpu().numpy()

    assert np.array_equal(run(kinds, nd), want)
    for ndecl in (nd - 1, nd + 1, 0, n):
        assert run(kinds, ndecl).sum() == 0
    k3 = kinds.copy()
    k3[int(np.nonzero(kinds == 0)[0][3])] = 9          # unknown kind: the single count drops by one
    k3[int(np.nonzero(kinds == 1)[0][5])] = 200
    assert run(k3, nd).sum() == 0
    assert run(k3, nd - 1).sum() == 0                   # nd - 1 doubles now, but the singles are short too
    # the pieces, used directly with an understated / overstated count and a poisoned index vector
    idx = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    idx2 = torch.full((n,), -1, dtype=torch.int32, device="cuda:0")
    scratch = torch.empty(engine.split_scratch_bytes(n), dtype=torch.uint8, device="cuda:0")
    engine.split_kinds_dev(_dev(kinds), idx, idx2, scratch)
    cnt = engine.split_counts(scratch)
    dst = torch.zeros((n, 64), dtype=torch.uint8, device="cuda:0")
    engine.gather_rows_dev(t["R"], idx, n, dst, limit=cnt[0:1])       # count = n > what the split wrote
    out = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    engine.scatter_verdicts_dev(torch.ones(n, dtype=torch.uint8, device="cuda:0"), idx, n, out, limit=cnt[0:1])
    engine.gather_rows_dev(t["R"], idx, n, dst)                        # no limit: bad indices are skipped
    torch.cuda.synchronize()
    ns = n - nd
    assert np.array_equal(dst[:ns].cpu().numpy(), cols["R"][kinds == 0])
    assert np.array_equal(out.cpu().numpy(), (kinds == 0).astype(np.uint8))


def test_mixed_sharded_verifier_refuses_wrong_counts(engine):
    """MixedShardedVerifier with a declared kind count that disagrees with the
    kind vector returns all zeros (decided on the device) instead of a silently wrong mapping."""
    import torch
    from schnorr_amd.distributed import MixedShardedVerifier
    n = 2048
    kinds, cols, want, nd = _mixed_arrays(n, 29)
    batch = {k: _dev(v) for k, v in cols.items()}
    batch["kinds"] = _dev(kinds)
    good = MixedShardedVerifier(n, nd, 1, 0, "cuda:0", collective=False)
    out = good(batch, batch["kinds"]).cpu().numpy()
    assert np.array_equal(out, want) and good.local_counts() == (n - nd, nd)
    for wrong in (nd - 1, nd + 3):
        bad = MixedShardedVerifier(n, wrong, 1, 0, "cuda:0", collective=False)
        assert int(bad(batch, batch["kinds"]).sum()) == 0
        assert bad.local_counts() == (n - nd, nd)


def test_half_gcd_near_equal_remainders_do_not_spin(engine):
    """8r = 41 c + e with tiny e makes the first remainder c + e, i.e. X / Y within
    2^-30 of 1 — r02's estimate-only loop ran to its iteration cap on such a lane.  The challenge is
    handed to the second-stage entry point directly (a hash cannot be steered there)."""
    import torch
    N8 = 8 * M.R_ORDER
    cs = []
    for k in (1, 2, 5):
        e = N8 % 41 + 41 * k
        c = (N8 - e) // 41
        assert (N8 - e) % 41 == 0 and c < (1 << 250) and N8 - 40 * c == c + e
        cs.append(c)
    cs.append(cs[0] + 1)                                   # a neighbour that behaves normally
    n = len(cs)
    rng = np.random.default_rng(12)
    sk = [int.from_bytes(rng.bytes(31), "little") % M.R_ORDER for _ in range(n)]
    rr = [int.from_bytes(rng.bytes(31), "little") % M.R_ORDER for _ in range(n)]
    le = lambda xs: np.frombuffer(b"".join(M.le32(x) for x in xs), np.uint8).reshape(len(xs), 32).copy()
    PK = engine.public_keys(le(sk))
    R = engine.public_keys(le(rr))
    u = le([(rr[i] - cs[i] * sk[i]) % M.R_ORDER for i in range(n)])
    for tampered in (False, True):
        uu = u.copy()
        if tampered:
            uu[:, 0] ^= 1
        ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        ws = torch.empty(engine.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
        valid = torch.ones(n, dtype=torch.uint8, device="cuda:0")
        os.environ.pop("DSV_QUAD", None)
        engine.verify_core_dev(_dev(uu), _dev(le(cs)), valid, _dev(PK), _dev(R), ok, ws)
        torch.cuda.synchronize()
        assert ok.cpu().numpy().tolist() == [0 if tampered else 1] * n
    # the same scalars through the oracle's plain equation u*G + c*PK == R
    for i in range(n):
        lhs = M.padd(M.pmul(M.GEN, int.from_bytes(u[i].tobytes(), "little")),
                     M.pmul(H.to_int_point(PK[i]), cs[i]))
        assert lhs == H.to_int_point(R[i])


@pytest.mark.parametrize("world,config", [(4, "single"), (5, "single"), (5, "mixed")])
def test_bench_rehearses_larger_world_sizes_on_one_gpu(world, config):
    """`python bench.py --gpus N` spawns N fresh ranks (gloo, all on GPU 0): ports,
    shard arithmetic, N contexts' memory, gathered verdicts of every other rank.  The default config
    at N > 1 also runs configs[4] (mixed) in the same process group.  The GPU box admits six
    processes on its card and this test process is one of them, so N = 5 is the largest rehearsal
    that may run here; world size 8 is covered on the CPU (tests/test_distributed.py)."""
    drop = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update({"DSV_BENCH_DEVICE": "0", "DSV_BENCH_BACKEND": "gloo"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1",
           "--warmup", "1", "--log2-batch", "13", "--config", config, "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["world_size"] == world and line["backend"] == "gloo"
    assert line["launcher"] == "self-spawned" and line["value"] > 0
    if config == "single":
        assert line["mixed"]["n_gpus"] == world
    # r04: with N > 1 the line says which rank lags and where its time goes
    def per_rank(d, unit_positive=True):
        assert set(d) >= {"min", "max", "argmax", "all"} and len(d["all"]) == world
        assert 0 <= d["argmax"] < world and d["min"] <= d["max"]
        assert abs(d["all"][d["argmax"]] - d["max"]) < 1e-3
        if unit_positive:
            assert d["min"] > 0
    ranks = line["ranks"]
    per_rank(ranks["ms_per_step"])
    per_rank(ranks["init_s"])
    assert abs(ranks["ms_per_step"]["max"] - line["ms_per_step"]) < 1e-6 * max(1.0, line["ms_per_step"])
    if config == "single":
        per_rank(ranks["verify_ms"])
        per_rank(ranks["all_gather_ms"], unit_positive=False)
        mixed = line["mixed"]["ranks"]
    else:
        mixed = ranks
    stages = ("split_gather_ms", "single_kernels_ms", "double_kernels_ms", "all_gather_ms", "scatter_ms")
    for k in stages:
        per_rank(mixed[k], unit_positive=False)
    slow = mixed["slowest_rank"]
    assert slow["rank"] == mixed["ms_per_step"]["argmax"] and all(k in slow for k in stages)
    assert slow["single_kernels_ms"] > 0 and slow["double_kernels_ms"] > 0


def test_device_half_scalars_are_those_of_the_exact_euclid(engine):
    """halfgcd.h on the device (quotient estimates from double-precision images, alternating roles)
    against the integer model: the pair is the first Euclidean remainder below 2^128 and its
    cofactor — EQUAL to tests/pymodel.py's half_scalars, which never touches a float — for random c,
    for c next to the thresholds, and for c crafted so that a remainder comes within one unit of
    its partner (the r02 spin).  Inputs that need more than the iteration cap (c = 2^128: quotient
    2^127, taken 31 bits at a time) still give a valid, longer pair."""
    import random
    rnd = random.Random(4242)
    N8 = 8 * M.R_ORDER
    edge = [0, 1, 2, 3, (1 << 128) - 1, (1 << 250) - 1, (1 << 250) - 2, N8 >> 6, N8 >> 5, (N8 // 3) >> 4,
            (N8 - ((N8 % 41) + 41)) // 41, (N8 - ((N8 % 1000003) + 1000003)) // 1000003,
            (1 << 249) + 1]
    # a first quotient far above 2^31 may hit the iteration cap: valid, but not Euclid's final pair
    slow = [1 << 128, (1 << 128) + 1, (1 << 200) + 12345, (1 << 129) + 1, (1 << 160) - 1] + \
        [rnd.getrandbits(rnd.randrange(129, 226)) for _ in range(500)]
    for _ in range(64):   # c ~ N / k: the first quotient is k, the remainder small
        k = rnd.randrange(33, 1 << rnd.randrange(6, 29))
        edge.append(N8 // k)
    for _ in range(64):   # two steps from the end, a remainder almost equal to its partner
        e = rnd.randrange(1, 1 << 20)
        edge.append(((N8 - e) // 2) & ((1 << 250) - 1))
    cases = edge + slow + [rnd.getrandbits(250) for _ in range(20000)] + \
        [rnd.getrandbits(rnd.randrange(226, 251)) for _ in range(2000)]
    le = np.frombuffer(b"".join(M.le32(c) for c in cases), np.uint8).reshape(len(cases), 32).copy()
    got = engine.debug_half_scalars(le)
    slow_set = set(slow)
    for c, (a, b) in zip(cases, got):
        assert (a - b * c) % N8 == 0 and b & 1 and a >= 0, c
        assert 0 < abs(b) < (1 << 160) and a < (1 << 251), c
        if c not in slow_set:
            ma, mb, mneg = M.half_scalars(c)
            assert (a, b) == (ma, -mb if mneg else mb), c


def test_device_lattice_scalars_satisfy_the_congruences(engine):
    """lattice3.h on the device: x = z*u, y = z*c (mod 8r), z odd, ~170 bits; degenerate inputs fall
    back to (u, c, 1) or still give a valid triple.  (The floating-point decisions may differ from
    the Python model's: only the congruences, the parity and the size are properties.)"""
    import random
    rnd = random.Random(77)
    N8 = 8 * M.R_ORDER
    edge = [(0, 0), (1, 1), (0, 5), (M.R_ORDER - 1, (1 << 250) - 1), (12345, 0), (M.R_ORDER - 1, 1),
            (1 << 200, 1 << 100), (M.R_ORDER - 1, (1 << 250) - 2), (7, (1 << 250) - 3), (1, 0), (0, 1)]
    cases = edge + [(rnd.randrange(M.R_ORDER), rnd.getrandbits(250)) for _ in range(2000)]
    le = lambda xs: np.frombuffer(b"".join(M.le32(x) for x in xs), np.uint8).reshape(len(xs), 32).copy()
    got = engine.debug_lattice3(le([u for u, _ in cases]), le([c for _, c in cases]))
    sizes = []
    for (u, c), (x, y, z) in zip(cases, got):
        assert (x - z * u) % N8 == 0 and (y - z * c) % N8 == 0, (u, c)
        assert z & 1 and 0 < abs(z) < M.R_ORDER
        assert max(abs(x), abs(y), abs(z)) < (1 << 252)
        sizes.append(max(abs(v).bit_length() for v in (x, y, z)))
    rand = sorted(sizes[len(edge):])
    assert rand[len(rand) // 2] <= 171 and rand[int(0.999 * len(rand))] <= 176, (rand[len(rand) // 2], rand[-1])


def test_vargen_order8_torsion_components_valid_and_invalid(engine):
    """Generators, keys and nonce points carrying an order-8 component through the three-scalar
    kernel: valid exactly when the torsion parts cancel — the reference equation's verdicts (oracle)."""
    import test_halfgcd as TH
    t8 = TH.order8_point()
    rnd = TH.rnd
    rows = {"u": [], "R": [], "PK": [], "Gen": [], "m": []}
    want = []
    while sum(want) < 3 or len(want) < 48:
        sk, m, rr, g = (rnd.randrange(1, M.R_ORDER), rnd.randrange(M.Q), rnd.randrange(1, M.R_ORDER),
                        rnd.randrange(1, M.R_ORDER))
        k0, k1 = rnd.randrange(8), rnd.randrange(1, 8)
        gen0 = M.pmul(M.GEN, g)
        gen = M.padd(gen0, M.pmul(t8, k0))
        pk = M.padd(M.pmul(gen0, sk), M.pmul(t8, k1))
        for k2 in range(8):
            R = M.padd(M.pmul(gen0, rr), M.pmul(t8, k2))
            c = M.challenge(R, m)
            u = (rr - c * sk) % M.R_ORDER
            rows["u"].append(np.frombuffer(M.le32(u), np.uint8))
            rows["R"].append(np.frombuffer(M.point_bytes(R), np.uint8))
            rows["PK"].append(np.frombuffer(M.point_bytes(pk), np.uint8))
            rows["Gen"].append(np.frombuffer(M.point_bytes(gen), np.uint8))
            rows["m"].append(np.frombuffer(M.le32(m), np.uint8))
            want.append(int((u * k0 + c * k1 - k2) % 8 == 0))
    a = {k: np.stack(v) for k, v in rows.items()}
    cpu = O.verify_vargen(a["u"], a["R"], a["PK"], a["Gen"], a["m"], nthreads=8)
    assert list(cpu) == want
    got = engine.verify_vargen(a["u"], a["R"], a["PK"], a["Gen"], a["m"])
    assert list(got) == want
    # identity / small-order generator and key, default signature: complete formulas, no special case
    ident = np.frombuffer(M.point_bytes(M.IDENTITY), np.uint8)
    z32 = np.zeros(32, np.uint8)
    small = np.frombuffer(M.point_bytes(t8), np.uint8)
    spec = {"u": np.stack([z32, z32, a["u"][0]]), "R": np.stack([ident, small, a["R"][0]]),
            "PK": np.stack([ident, small, ident]), "Gen": np.stack([ident, ident, small]),
            "m": np.stack([a["m"][0]] * 3)}
    assert np.array_equal(engine.verify_vargen(*(spec[k] for k in ("u", "R", "PK", "Gen", "m"))),
                          O.verify_vargen(*(spec[k] for k in ("u", "R", "PK", "Gen", "m"))))
