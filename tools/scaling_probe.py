"""Kernel time vs batch size: T(n) = a + b n for the two kernels of a single verification."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from schnorr_amd import engine as E, workload as W
E.init(0)
dev = "cuda:0"
nmax = 1 << 22
b = W.gen_single(nmax, 2321, tamper=False)
ws = torch.empty(E.workspace_bytes(nmax), dtype=torch.uint8, device=dev)
for lg in (16, 17, 18, 19, 20, 21, 22):
    n = 1 << lg
    sl = {k: b[k][:n] for k in ("u", "R", "PK", "m")}
    c = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    valid = torch.empty(n, dtype=torch.uint8, device=dev)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    for rep in range(2):
        E.challenge_single_dev(sl["R"], sl["m"], c, valid)
        E.verify_core_dev(sl["u"], c, valid, sl["PK"], sl["R"], ok, ws)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    reps = 4
    th = tv = 0.0
    for rep in range(reps):
        ev[0].record(); E.challenge_single_dev(sl["R"], sl["m"], c, valid)
        ev[1].record(); E.verify_core_dev(sl["u"], c, valid, sl["PK"], sl["R"], ok, ws)
        ev[2].record(); torch.cuda.synchronize()
        th += ev[0].elapsed_time(ev[1]); tv += ev[1].elapsed_time(ev[2])
    assert bool(ok.all())
    print("2^%d  hash %.3f ms (%.3f per 2^20)   verify %.3f ms (%.3f per 2^20)"
          % (lg, th / reps, th / reps * (1 << 20) / n, tv / reps, tv / reps * (1 << 20) / n))
