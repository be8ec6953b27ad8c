#!/usr/bin/env python3
"""bench.py — Schnorr verifies/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2-batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the verify hot path over one batch of 2^20 synthetic single signatures
per GPU (BASELINE.json configs[1]), inputs already resident in HBM, through the public
dsv_verify_single_dev: k_challenge (Poseidon) then k_verify_fixed_half ((b*u)*G + a*PK - b*R == O,
halfgcd.h), which the library cuts into 2^16-item sub-batches on two internal streams.  Batches are generated on the GPU by the engine's own sign
kernels and every 16th item is corrupted, so the expected verdict vector is non-trivial; it is
checked after the timed region (and a sample is re-verified by the CPU oracle at N = 1).

N > 1: one process per GPU, each rank verifies its own 2^20-item shard (weak scaling) and the
verdict bytes are all-gathered over RCCL inside the timed region.

Prints ONE JSON line on rank 0.  `roofline` covers both kernels of the step (each is VALU-issue
bound, and their sub-batches overlap, so no single launch can be bracketed inside the timed
region): lane-instructions per step / step time — see DESIGN.md §4 for the instruction model; the
single-launch durations of each kernel, taken after the timed region, and HBM figures ride along.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- work model of the dominant kernel k_verify_fixed_half --------------------
# VALU lane-instructions per verdict, from the instruction counts hipcc emits for each field
# operation (tools: `hipcc -S` of fe29.h; checked against rocprof SQ_INSTS_VALU in profiles/).
FE_MUL, FE_SQR = 211, 181            # 153 / 117 v_mad_u64_u32 + carry & normalise ops
ADD, SUB, CARRY = 9, 45, 26          # limb-wise add; biased subtract + carry pass; carry pass
SUB_RAW = SUB - CARRY                # biased subtract whose consumers tolerate un-carried limbs
DOUBLE = 3 * FE_SQR + 4 * FE_MUL + 3 * ADD + SUB + SUB_RAW    # uu, vv, zz; 2uv, and the 3 outputs
ADD_NIELS = 8 * FE_MUL + 4 * ADD + CARRY + SUB + 2 * SUB_RAW
ADD_ANIELS = 7 * FE_MUL + 4 * ADD + CARRY + SUB + 2 * SUB_RAW
TO_NIELS = 2 * FE_MUL + ADD + CARRY + 2 * SUB          # incl. the negated 2d*t of a table entry
TABLE9 = 7 * ADD_NIELS + 8 * TO_NIELS                  # |d|*P, d = 1..8
WINDOWS = 33                                           # mean over waves of the longest lane's digits
HALF_GCD = 13000                                       # ~90 iterations x ~140 instructions
VERIFY_INSTR = (
    4 * FE_MUL                                         # PK, R to Montgomery form
    + 2 * TABLE9                                       # window tables of PK and R
    + HALF_GCD + 2 * 8 * 90                            # (a, b) and b*u mod r
    + WINDOWS * (4 * DOUBLE + 2 * ADD_NIELS)           # a*PK -/+ b*R, shared doublings
    + 23 * (ADD_ANIELS + 12)                           # += (b*u)*G, signed 11-bit windows
    + 400                                              # identity test
)
# k_challenge (hades29.h): 8 full rounds (5 S-boxes + five 5-term constant dots), 59 partial
# rounds in blocks of 4 (4 S-boxes, rows of 5..8 terms, four 5-term updates), last block of 3
SBOX = 2 * FE_SQR + FE_MUL
DOT = lambda nt: 81 * nt + (FE_MUL - 81)               # nt limb products, one reduction
HASH_INSTR = (
    3 * FE_MUL                                         # Ru, Rv, m to Montgomery form
    + 8 * (5 * SBOX + 5 * DOT(5))
    + 14 * (4 * SBOX + DOT(5) + DOT(6) + DOT(7) + DOT(8) + 4 * DOT(5))
    + (3 * SBOX + DOT(5) + DOT(6) + DOT(7) + 4 * DOT(5))
    + FE_MUL + 150                                     # out of Montgomery form, truncate, store
)
ALGO_BYTES_SINGLE = 193                            # 192 B in + 1 B out
CORE_BYTES = 32 + 32 + 1 + 64 + 64 + 1             # what k_verify_fixed_half moves per item
HASH_BYTES = 64 + 32 + 32 + 1                      # k_challenge: R, m in; c, valid out
VALU_CYCLES_PER_INSTR = 4.05                       # measured: profiles/r01_valu_rates.txt
N_CU, SIMD_PER_CU, CLOCK_HZ = 256, 4, 2.4e9
VALU_PEAK_LANE_INSTR = N_CU * SIMD_PER_CU * CLOCK_HZ / VALU_CYCLES_PER_INSTR * 64
HBM_PEAK_GBS = 8000.0


def _pmc_traffic(n):
    """HBM bytes per step (k_verify_fixed_half + k_challenge) from the last committed rocprofv3 PMC pass
    (profiles/pmc_latest.json; FETCH_SIZE + WRITE_SIZE, KB -> bytes, scaled to this batch).
    Counters cannot be read from inside the timed process, so this is the recorded figure, not a
    live one; null when no profile is committed.  (gfx950 under-reports FETCH_SIZE by up to 2x
    for wide coalesced reads; these are 16-B-per-lane scattered reads, uncalibrated.)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            p = json.load(f)
        kb = p["FETCH_SIZE_KB"] + p["WRITE_SIZE_KB"]
        h = p.get("k_challenge", {})                       # the hash kernel of the same step
        kb += h.get("FETCH_SIZE_KB", 0.0) + h.get("WRITE_SIZE_KB", 0.0)
        return kb * 1024.0 * n / p["batch"]
    except Exception:
        return None


def _pmc_valu_busy():
    """VALU issue-slot occupancy of the dominant kernel in the last committed PMC pass:
    SQ_INSTS_VALU wave-instructions x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            p = json.load(f)
        return p["SQ_INSTS_VALU"] * 4.0 / (N_CU * SIMD_PER_CU * p["GRBM_GUI_ACTIVE"] / 8.0)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-batch", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-double", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    # rehearsal knobs (one-GPU boxes): DSV_BENCH_DEVICE pins every rank to one GPU and
    # DSV_BENCH_BACKEND=gloo replaces RCCL, so the N > 1 code path can be exercised without N GPUs
    dev_index = int(os.environ.get("DSV_BENCH_DEVICE", local_rank))
    backend = os.environ.get("DSV_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index

    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    from schnorr_amd import engine as E
    from schnorr_amd import workload as W

    E.init(dev_index)
    n = 1 << args.log2_batch
    batch = W.gen_single(n, seed=2321, device=dev, first_item=rank * n)  # one stream, sharded
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    ws = torch.empty(E.workspace_bytes(n), dtype=torch.uint8, device=dev)
    c = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    valid = torch.empty(n, dtype=torch.uint8, device=dev)
    gathered = torch.empty(world * n, dtype=torch.uint8, device=dev) if world > 1 else None

    def step():
        # the public device-pointer entry point: k_challenge + k_verify_fixed_half, cut by the
        # library into 2^16-signature sub-batches on two internal streams (forked from / joined to
        # the current stream)
        E.verify_single_dev(batch["u"], batch["R"], batch["PK"], batch["m"], ok, ws)
        if world > 1:
            dist.all_gather_into_tensor(gathered, ok)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # communicator set-up (RCCL connects lazily on the first collective) is not a step
        dist.all_gather_into_tensor(gathered, ok)
    sync_all()
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    ok_api = ok.clone()

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # ---- per-kernel durations for the roofline: the same two kernels over the whole batch, one
    # launch each on the current stream, bracketed by events (outside the timed region; inside it
    # the sub-batches of the two kernels overlap, so no single launch can be bracketed)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(3)]
    for e in evs:
        e[0].record()
        E.challenge_single_dev(batch["R"], batch["m"], c, valid)
        e[1].record()
        E.verify_core_dev(batch["u"], c, valid, batch["PK"], batch["R"], ok, ws)
        e[2].record()
    torch.cuda.synchronize()
    hash_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs)
    core_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs)

    # ---- correctness of what was timed
    mism = int((ok != batch["expected"]).sum().item())
    mism_api = int((ok_api != batch["expected"]).sum().item())
    if world > 1:
        mine = gathered[rank * n:(rank + 1) * n]
        mism += int((mine != batch["expected"]).sum().item())
    if mism or mism_api:
        raise SystemExit("rank %d: %d / %d verdicts differ from the expected pattern"
                         % (rank, mism, mism_api))

    total = n * world
    value = total * args.steps / dt
    out = {
        "metric": "Schnorr verifies/sec (single) at batch=2^%d" % args.log2_batch,
        "value": value,
        "unit": "verifies/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32x9 (29-bit limbs, u64 accumulate)",
        "data": "synthetic: reference harness inputs (StdRng::seed_from_u64(2321): sk, m, nonce per "
                "item, restated RNG), GPU-signed, every 16th item corrupted",
        "config": {"workload": "2^%d single-signature batch verify per GPU (BASELINE configs[1])"
                               % args.log2_batch,
                   "batch_per_gpu": n, "parallelism": "dp%d" % world,
                   "collective": "all_gather of verdict bytes" if world > 1 else "none"},
    }

    if rank == 0:
        # The timed region IS the bracket: the library forks its two internal streams from the
        # current stream and joins them back, so `dt` covers exactly steps x (k_challenge +
        # k_verify_fixed_half over n items).  Both kernels are VALU-issue bound; achieved = the
        # lane-instructions of both per step / the step time of this rank's own launches.
        step_s = dt / args.steps
        core_s = core_ms * 1e-3
        lane_instr = (VERIFY_INSTR + HASH_INSTR) * n
        achieved = lane_instr / step_s
        out["roofline"] = {
            "kernel": "k_challenge + k_verify_fixed_half (2^16-item sub-batches, two streams)",
            "bound": "valu",
            "achieved": achieved / 1e12,
            "peak": VALU_PEAK_LANE_INSTR / 1e12,
            "unit": "T lane-instr/s",
            "frac": achieved / VALU_PEAK_LANE_INSTR,
            "traffic": _pmc_traffic(n),
            "valu_busy_from_pmc": _pmc_valu_busy(),
            "model": {"valu_lane_instr_per_verdict": {"k_verify_fixed_half": VERIFY_INSTR,
                                                      "k_challenge": HASH_INSTR},
                      "cycles_per_wave_instr": VALU_CYCLES_PER_INSTR,
                      # one full-batch launch of each kernel on one stream, after the timed region
                      "kernel_ms": core_ms, "hash_kernel_ms": hash_ms,
                      "verify_kernel_frac_alone": VERIFY_INSTR * n / core_s / VALU_PEAK_LANE_INSTR,
                      "hash_kernel_frac_alone": HASH_INSTR * n / (hash_ms * 1e-3)
                                                / VALU_PEAK_LANE_INSTR},
            "hbm": {"bound": "hbm", "achieved": (CORE_BYTES + HASH_BYTES) * n / step_s / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (CORE_BYTES + HASH_BYTES) * n / step_s / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_verdict": ALGO_BYTES_SINGLE,
                    "kernel_bytes_per_verdict": CORE_BYTES + HASH_BYTES},
        }

    # ---- secondary figure: double signatures (BASELINE configs[2]), outside the timed region
    if not args.no_double and world == 1:
        nd = n
        bd = W.gen_double(nd, seed=4242, device=dev)
        okd = torch.zeros(nd, dtype=torch.uint8, device=dev)
        E.verify_double_dev(bd["u"], bd["R"], bd["Rp"], bd["PK"], bd["PKp"], bd["m"], okd, ws)
        torch.cuda.synchronize()
        td0 = time.perf_counter()
        reps = max(1, args.steps // 2)
        for _ in range(reps):
            E.verify_double_dev(bd["u"], bd["R"], bd["Rp"], bd["PK"], bd["PKp"], bd["m"], okd, ws)
        torch.cuda.synchronize()
        tdd = time.perf_counter() - td0
        if int((okd != bd["expected"]).sum().item()):
            raise SystemExit("double-signature verdicts differ from the expected pattern")
        out["double"] = {"value": nd * reps / tdd, "unit": "verifies/s",
                         "workload": "2^%d double-signature batch (BASELINE configs[2])"
                                     % args.log2_batch}
        del bd, okd
        # var-generator scheme (BASELINE configs[3]: 2^18), both bases variable
        nv = min(n, 1 << 18)
        bv = W.gen_vargen(nv, seed=777, device=dev)
        okv = torch.zeros(nv, dtype=torch.uint8, device=dev)
        E.verify_vargen_dev(bv["u"], bv["R"], bv["PK"], bv["Gen"], bv["m"], okv, ws)
        torch.cuda.synchronize()
        tv0 = time.perf_counter()
        for _ in range(reps):
            E.verify_vargen_dev(bv["u"], bv["R"], bv["PK"], bv["Gen"], bv["m"], okv, ws)
        torch.cuda.synchronize()
        tvv = time.perf_counter() - tv0
        if int((okv != bv["expected"]).sum().item()):
            raise SystemExit("var-generator verdicts differ from the expected pattern")
        # signing (the step in front of verify): R = r*G, c = H(R, m), u = r - c*sk
        sk_ = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev); sk_[:, 31] &= 0x07
        r_ = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev); r_[:, 31] &= 0x07
        su = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        sR = torch.empty((n, 64), dtype=torch.uint8, device=dev)
        E.sign_single_dev(sk_, batch["m"], r_, su, sR)
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(reps):
            E.sign_single_dev(sk_, batch["m"], r_, su, sR)
        torch.cuda.synchronize()
        out["sign"] = {"value": n * reps / (time.perf_counter() - ts0), "unit": "signatures/s",
                       "workload": "2^%d single signatures, nonces supplied" % args.log2_batch}
        del sk_, r_, su, sR
        out["vargen"] = {"value": nv * reps / tvv, "unit": "verifies/s",
                         "workload": "2^%d var-generator batch (BASELINE configs[3])"
                                     % (nv.bit_length() - 1)}
        del bv, okv

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O

        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, 16))  # the GPU box's CPU share for one GPU
        sample = 4096 * cores
        hu = batch["u"][:sample].cpu().numpy()
        hR = batch["R"][:sample].cpu().numpy()
        hPK = batch["PK"][:sample].cpu().numpy()
        hm = batch["m"][:sample].cpu().numpy()
        O.verify_single(hu[:64], hR[:64], hPK[:64], hm[:64])  # warm
        tc0 = time.perf_counter()
        cpu_ok = O.verify_single(hu, hR, hPK, hm, nthreads=cores)
        tc = time.perf_counter() - tc0
        want = batch["expected"][:sample].cpu().numpy()
        if (cpu_ok != want).any() or (cpu_ok != ok[:sample].cpu().numpy()).any():
            raise SystemExit("CPU oracle disagrees with the GPU verdicts on the sample")
        one = min(sample, 2048)
        t10 = time.perf_counter()
        O.verify_single(hu[:one], hR[:one], hPK[:one], hm[:one], nthreads=1)
        t1 = time.perf_counter() - t10
        ts = time.perf_counter()
        O.keygen_sign_single(1024, 0xBEEF, nthreads=1)  # BASELINE configs[0] shape: keygen + sign
        t_sign = time.perf_counter() - ts
        out["cpu_baseline"] = {
            "value": sample / tc, "unit": "verifies/s", "cores": cores, "kind": "port",
            "sample": "first %d items of the same batch, %d threads, %.1f s wall; "
                      "1 thread: %.0f verifies/s on %d items; configs[0] shape (1024 x keygen+sign, "
                      "1 thread): %.0f /s" % (sample, cores, tc, one / t1, one, 1024 / t_sign),
        }
        # host-buffer path of the C ABI (PCIe-inclusive), never the headline value
        hs = n
        hu = batch["u"][:hs].cpu().numpy(); hR = batch["R"][:hs].cpu().numpy()
        hPK = batch["PK"][:hs].cpu().numpy(); hm = batch["m"][:hs].cpu().numpy()
        E.verify_single(hu, hR, hPK, hm)  # warm: staging buffers sized for this batch
        th0 = time.perf_counter()
        E.verify_single(hu, hR, hPK, hm)
        th = time.perf_counter() - th0
        out["host_path"] = {"value": hs / th, "unit": "verifies/s",
                            "note": "dsv_verify_single on %d host-resident items incl. PCIe staging" % hs}
        # BASELINE configs[0] size through the same host entry point: latency of a 1024-item call
        E.verify_single(hu[:1024], hR[:1024], hPK[:1024], hm[:1024])
        tl0 = time.perf_counter()
        for _ in range(20):
            E.verify_single(hu[:1024], hR[:1024], hPK[:1024], hm[:1024])
        tl = (time.perf_counter() - tl0) / 20
        out["small_batch"] = {"ms_per_call": tl * 1e3, "value": 1024 / tl, "unit": "verifies/s",
                              "workload": "1024 single signatures per call, host buffers "
                                          "(BASELINE configs[0] size)"}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
