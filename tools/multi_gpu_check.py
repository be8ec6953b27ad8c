#!/usr/bin/env python3
"""First-real-multi-GPU checklist: ONE command that exercises every branch a one-GPU box cannot reach.

    python tools/multi_gpu_check.py [--gpus N] [--log2-batch 18] [--skip-bench]

Everything below has been built and rehearsed on one GPU (contiguous shards wrapped over one device,
gloo world sizes 2 .. 8 on the CPU) but has never seen a device ordinal other than 0 nor RCCL with more
than one rank.  On a node with N >= 2 visible GPUs this script runs, in order, and prints
PASS / FAIL per step with the figures that matter; it stops at the first FAIL (no GPU step is started
after a failed one):

 1. dsv_init_visible: a context (tables, streams) on every device; per-device init time.
 2. host entry points ON device d != 0 (dsv_set_device(d)): affine, projective, limb and wire forms —
    verdicts against the expected pattern of a GPU-signed, tampered batch (generated on device 0).
 3. device-pointer entry points with buffers OWNED by device d != 0 while the calling thread's current
    device is 0 (pointer-owner routing + DeviceGuard): verdicts, and the current device is unchanged.
 4. dsv_verify_*_multi and the column entry points (what verify_batch binds) sharded over ALL devices:
    verdicts, wall time against the one-device call (per-GPU efficiency of the HOST path).
 5. two batches in flight (submit / wait) sharded over all devices.
 6. `bench.py --gpus N` for the single and the mixed configuration through torch.distributed.run (RCCL
    all_gather of the verdict shards): the JSON lines, per-rank step times, time inside all_gather.
With one visible GPU it says so and runs steps 1, 2 (device 0), 4 and 5 with DSV_MULTI_SHARDS=3 (the
sharding arithmetic only — that is the rehearsal already on record, not a multi-GPU result).

The reference path this shards: PublicKey::verify /root/reference/src/keys/public.rs:121-130 (every
signature independent: contiguous shards, no collective inside the data path).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def step(name, ok, detail=""):
    print("[%s] %s%s" % ("PASS" if ok else "FAIL", name, (": " + detail) if detail else ""), flush=True)
    if not ok:
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="0 = every visible device")
    ap.add_argument("--log2-batch", type=int, default=18)
    ap.add_argument("--skip-bench", action="store_true")
    args = ap.parse_args()
    import torch

    from schnorr_amd import engine as E
    from schnorr_amd import workload as W

    visible = torch.cuda.device_count()
    ngpu = min(args.gpus or visible, visible)
    step("visible devices", visible >= 1, "%d visible, using %d" % (visible, ngpu))
    rehearsal = ngpu < 2
    if rehearsal:
        print("[NOTE] one GPU: steps 3 and 6 need >= 2; the rest runs as a sharding rehearsal (DSV_MULTI_SHARDS=3)")
        os.environ["DSV_MULTI_SHARDS"] = "3"
    else:
        os.environ["DSV_DEVICES"] = ",".join(str(d) for d in range(ngpu))

    # ---- 1. contexts on every device
    times = []
    for d in range(ngpu):
        t0 = time.perf_counter()
        E.init(d)
        times.append(time.perf_counter() - t0)
    step("1. dsv_init on every device", sorted(E.initialized_devices()) == list(range(ngpu)),
         "init ms per device: " + " ".join("%.0f" % (t * 1e3) for t in times))
    # ---- 1b. where each device's host-side threads run (include/dsv.h: dsv_device_numa)
    nodes = set()
    for d in range(ngpu):
        info = E.device_numa(d)
        cpus = info["cpus"]
        span = ("%d cpus %d..%d" % (len(cpus), cpus[0], cpus[-1])) if cpus else "no binding"
        print("       device %d  pci %s  numa node %d  copy threads: %s" % (d, info["bdf"], info["node"], span), flush=True)
        nodes.add(info["node"])
    step("1b. NUMA placement of every device", True,
         "nodes in use: %s%s" % (sorted(nodes), " (unknown: the copy threads float; a container that hides /sys/bus/pci)"
                                 if -1 in nodes else ""))

    n = 1 << args.log2_batch
    b = W.gen_single(n, seed=2321, device="cuda:0")
    want = b["expected"].cpu().numpy()
    h = {k: b[k].cpu().numpy() for k in ("u", "R", "PK", "m")}
    z = np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8)
    z[:, 31] = 0
    z[:, 0] |= 1
    proj = lambda a: np.concatenate([E.debug_fq_mul(np.ascontiguousarray(a[:, :32]), z),
                                     E.debug_fq_mul(np.ascontiguousarray(a[:, 32:]), z), z], axis=1)
    R3, PK3 = proj(h["R"]), proj(h["PK"])
    sig = np.ascontiguousarray(np.concatenate([h["u"], E.compress_points(h["R"])], axis=1))
    pk = E.compress_points(h["PK"])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # limb form of the same values (test infrastructure: conversion only)
    mont = [O.to_mont(h["u"], fr=True), O.to_mont(R3), O.to_mont(PK3), O.to_mont(h["m"])]

    # ---- 2. host entry points on every device
    for d in range(ngpu):
        E.set_device(d)
        res = {"affine": E.verify_single(h["u"], h["R"], h["PK"], h["m"]),
               "projective": E.verify_single_ext(h["u"], R3, PK3, h["m"]),
               "limbs": E.verify_single_mont(*mont),
               "wire": E.verify_single_wire(sig, pk, h["m"])}
        bad = [k for k, v in res.items() if not np.array_equal(v, want)]
        step("2. host entry points on device %d" % d, not bad, "differ: %s" % bad if bad else "4 input forms, %d items" % n)
    E.set_device(0)

    # ---- 3. device-pointer entry points on buffers owned by device d != 0
    for d in range(1, ngpu):
        dev = "cuda:%d" % d
        t = {k: b[k].to(dev) for k in ("u", "R", "PK", "m")}
        ok = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        ws = torch.empty(E.workspace_bytes(n), dtype=torch.uint8, device=dev)
        torch.cuda.set_device(0)
        E.verify_single_dev(t["u"], t["R"], t["PK"], t["m"], ok, ws)
        cur = torch.cuda.current_device()
        torch.cuda.synchronize(dev)
        step("3. device-pointer call on buffers of device %d from a thread on device 0" % d,
             cur == 0 and np.array_equal(ok.cpu().numpy(), want), "current device afterwards: %d" % cur)

    # ---- 4. one host batch sharded over all devices
    def best(fn, reps=3):
        fn()
        t = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            got = fn()
            t = min(t, time.perf_counter() - t0)
        return t, got

    one_dev, _ = best(lambda: E.verify_single(h["u"], h["R"], h["PK"], h["m"]))
    for name, fn in (("dsv_verify_single_multi", lambda: E.verify_single_multi(h["u"], h["R"], h["PK"], h["m"])),
                     ("dsv_verify_single_ext_multi", lambda: E.verify_single_ext(h["u"], R3, PK3, h["m"], multi=True)),
                     ("dsv_verify_single_mont_cols (verify_batch)", lambda: E.verify_mont_cols("single", mont))):
        t, got = best(fn)
        step("4. %s over %d device(s)" % (name, ngpu), np.array_equal(got, want),
             "%.2f ms = %.1f M/s; one device, affine host path: %.2f ms -> x%.2f" % (
                 t * 1e3, n / t / 1e6, one_dev * 1e3, one_dev / t))

    # ---- 5. two batches in flight, sharded
    t0 = time.perf_counter()
    jobs = [E.submit_mont_cols("single", mont) for _ in range(2)]
    outs = [j.wait() for j in jobs]
    dt = time.perf_counter() - t0
    step("5. two batches in flight over %d device(s)" % ngpu, all(np.array_equal(o, want) for o in outs),
         "%.2f ms for both = %.1f M/s" % (dt * 1e3, 2 * n / dt / 1e6))

    # ---- 5b. the batch fast accept, one aggregate per device (the batch holds tampered items: every shard
    # falls back to its per-signature kernels; then the valid items alone: every shard's aggregate accepts)
    got, accepted = E.verify_mont_cols_rlc("single", mont)
    step("5b. dsv_verify_single_mont_cols_rlc over %d device(s), tampered batch" % ngpu,
         np.array_equal(got, want) and not accepted, "accepted=%d" % accepted)
    keep = np.flatnonzero(want)
    valid_cols = [np.ascontiguousarray(c[keep]) for c in mont]
    t, (got, accepted) = best(lambda: E.verify_mont_cols_rlc("single", valid_cols))
    step("5b. ... valid items alone", bool(got.all()) and accepted,
         "%d items, %.2f ms = %.1f M/s, accepted=%d" % (len(keep), t * 1e3, len(keep) / t / 1e6, accepted))
    del b

    # ---- 6. one process per GPU over RCCL
    if rehearsal or args.skip_bench:
        print("[SKIP] 6. bench.py --gpus N (needs >= 2 GPUs)" if rehearsal else "[SKIP] 6. bench.py (--skip-bench)")
        return 0
    E.shutdown()
    for config in ("single", "mixed"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpu),
               "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"),
               "--gpus", str(ngpu), "--steps", "10", "--warmup", "2", "--config", config, "--no-cpu-baseline"]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("DSV_DEVICES", None)
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
        line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
        ok = r.returncode == 0 and line is not None
        detail = r.stderr[-600:] if not ok else ""
        if ok:
            d = json.loads(line)
            detail = "%.1f M/s over %d GPUs, %.2f ms per step; %s" % (
                d["value"] / 1e6, d["n_gpus"], d["ms_per_step"],
                json.dumps({k: d[k] for k in ("per_rank", "gather", "rccl_version") if k in d})[:700])
        step("6. bench.py --gpus %d --config %s" % (ngpu, config), ok, detail)
    return 0


if __name__ == "__main__":
    sys.exit(main())
