#!/usr/bin/env python3
"""Large-batch soak of the device-pointer path: per seed a 2^20 single, a 2^19 double and a 2^18
var-generator batch are generated on the GPU (StdRng stream of that seed, every 16th item
corrupted), verified through dsv_verify_{single,double,vargen}_dev (sub-batch split, one-lane
kernels, the three-scalar lattice form for the var-generator scheme), compared with the
construction-time pattern everywhere and with the CPU oracle on a random sample.  One line per seed.

    python tools/soak_big.py [--seeds N] [--first-seed S] [--sample K]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import oracle_lib as O  # noqa: E402
from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=50)
    ap.add_argument("--first-seed", type=int, default=70000)
    ap.add_argument("--sample", type=int, default=2048)
    a = ap.parse_args()
    E.init(0)
    threads = min(16, len(os.sched_getaffinity(0)))
    n, nd = 1 << 20, 1 << 19
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    ws = torch.empty(E.workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
    rng = np.random.default_rng(1)
    total = sampled = 0
    t0 = time.time()
    for seed in range(a.first_seed, a.first_seed + a.seeds):
        b = W.gen_single(n, seed=seed)
        E.verify_single_dev(b["u"], b["R"], b["PK"], b["m"], ok, ws)
        torch.cuda.synchronize()
        assert torch.equal(ok, b["expected"]), ("single pattern", seed)
        idx = torch.from_numpy(np.sort(rng.choice(n, a.sample, replace=False))).to("cuda:0")
        sub = [b[k][idx].cpu().numpy() for k in ("u", "R", "PK", "m")]
        assert np.array_equal(O.verify_single(*sub, nthreads=threads), ok[idx].cpu().numpy()), ("single oracle", seed)
        d = W.gen_double(nd, seed=seed + 1000003)
        E.verify_double_dev(d["u"], d["R"], d["Rp"], d["PK"], d["PKp"], d["m"], ok[:nd], ws)
        torch.cuda.synchronize()
        assert torch.equal(ok[:nd], d["expected"]), ("double pattern", seed)
        idx = torch.from_numpy(np.sort(rng.choice(nd, a.sample // 2, replace=False))).to("cuda:0")
        sub = [d[k][idx].cpu().numpy() for k in ("u", "R", "Rp", "PK", "PKp", "m")]
        assert np.array_equal(O.verify_double(*sub, nthreads=threads), ok[:nd][idx].cpu().numpy()), ("double oracle", seed)
        nv = 1 << 18
        v = W.gen_vargen(nv, seed=seed + 2000003)
        E.verify_vargen_dev(v["u"], v["R"], v["PK"], v["Gen"], v["m"], ok[:nv], ws)
        torch.cuda.synchronize()
        assert torch.equal(ok[:nv], v["expected"]), ("vargen pattern", seed)
        idx = torch.from_numpy(np.sort(rng.choice(nv, a.sample // 2, replace=False))).to("cuda:0")
        sub = [v[k][idx].cpu().numpy() for k in ("u", "R", "PK", "Gen", "m")]
        assert np.array_equal(O.verify_vargen(*sub, nthreads=threads), ok[:nv][idx].cpu().numpy()), ("vargen oracle", seed)
        total += n + nd + nv
        sampled += a.sample + a.sample
        print("seed %d ok  (%d verdicts pattern-checked, %d oracle-checked, %.0f s)"
              % (seed, total, sampled, time.time() - t0), flush=True)
    print("BIG SOAK OK: %d seeds, %d verdicts pattern-checked, %d oracle-checked" % (a.seeds, total, sampled))


if __name__ == "__main__":
    main()
