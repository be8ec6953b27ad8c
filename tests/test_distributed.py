"""CPU tests of the multi-GPU host logic: shard arithmetic and a world_size-2 gloo run of the
shard -> verify -> all_gather path at world sizes 2 and 8 (verify_fn = the CPU oracle here; on the GPU box the same
code runs with the HIP engine and backend "nccl")."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import harness as H
import oracle_lib as O
from schnorr_amd import distributed as D


def test_shard_bounds_partition_everything():
    for n in (0, 1, 7, 64, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = D.shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_bounds(10, 2, 2)


def test_split_mixed():
    s, d = D.split_mixed([0, 1, 0, 1, 1, 0])
    assert list(s) == [0, 2, 5] and list(d) == [1, 3, 4]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = O.keygen_sign_single(n, 2321)
    H.tamper(d, period=5)
    full = D.verify_single_sharded(d["u"], d["R"], d["PK"], d["m"],
                                   verify_fn=lambda *a: O.verify_single(*a),
                                   to_tensor=lambda a: torch.from_numpy(a))
    want = O.verify_single(d["u"], d["R"], d["PK"], d["m"])
    q.put((rank, bool(np.array_equal(full.numpy(), want)), int(full.sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (2, 64), (8, 37), (8, 5)])  # ragged, even, world 8, empty shards
def test_gloo_gather_world_sizes_2_and_8(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert len({r[2] for r in res}) == 1 and res[0][2] > 0


def _mixed_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    kinds = (rng.integers(0, 3, size=n) == 0).astype(np.uint8)       # ~1/3 double, interleaved
    ns, nd = int((kinds == 0).sum()), int((kinds == 1).sum())
    s = O.keygen_sign_single(ns, 11)
    d = O.keygen_sign_double(nd, 12)
    H.tamper(s, period=4)
    d["PKp"][::3] = d["PKp"][1::3][: len(d["PKp"][::3])] if nd > 3 else d["PKp"][::3]
    single = (s["u"], s["R"], s["PK"], s["m"])
    double = (d["u"], d["R"], d["Rp"], d["PK"], d["PKp"], d["m"])
    full = D.verify_mixed_sharded(kinds, single, double,
                                  verify_single_fn=lambda *a: O.verify_single(*a),
                                  verify_double_fn=lambda *a: O.verify_double(*a),
                                  to_tensor=lambda a: torch.from_numpy(a))
    want = np.zeros(n, np.uint8)
    want[kinds == 0] = O.verify_single(*single)
    want[kinds == 1] = O.verify_double(*double)
    q.put((rank, bool(np.array_equal(full.numpy(), want)), int(full.sum()), int(n - full.sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 61), (8, 61), (8, 11)])
def test_gloo_mixed_batch_world_sizes_2_and_8(world, n):
    """BASELINE configs[4] shape at toy size: interleaved single / double signatures, each kind
    sharded on its own, verdicts gathered and scattered back into batch order.  World size 8 is the
    configuration's own (2^23 over 8 GPUs) — run here over gloo on the CPU, with ragged per-kind
    shards (61 items: 7-8 singles and 2-3 doubles per rank) and with shards that are EMPTY on some
    ranks (11 items: fewer doubles than ranks), since no 8-GPU node is available to the build."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert len({r[2] for r in res}) == 1 and res[0][2] > 0 and res[0][3] > 0   # same vector everywhere, both verdicts occur
