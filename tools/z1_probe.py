import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from schnorr_amd import engine as E
from schnorr_amd import workload as W
E.init(0)
n = 1 << 20
b = W.gen_single(n, seed=2321)
h = {k: b[k].cpu().numpy() for k in ("u", "R", "PK", "m")}
want = b["expected"].cpu().numpy()
z = np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8); z[:, 31] = 0; z[:, 0] |= 1
proj = lambda a: np.concatenate([E.debug_fq_mul(np.ascontiguousarray(a[:, :32]), z), E.debug_fq_mul(np.ascontiguousarray(a[:, 32:]), z), z], axis=1)
R3, PK3 = proj(h["R"]), proj(h["PK"])
one = np.zeros((n, 32), np.uint8); one[:, 0] = 1
z1 = lambda a: np.ascontiguousarray(np.concatenate([a, one], axis=1))
R1, PK1 = z1(h["R"]), z1(h["PK"])
def best(fn, reps=6):
    fn(); t = []
    for _ in range(reps):
        t0 = time.perf_counter(); got = fn(); t.append(time.perf_counter() - t0)
    assert (got == want).all()
    return min(t) * 1e3, sorted(t)[len(t)//2] * 1e3
def dev(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
okd = torch.zeros(n, dtype=torch.uint8, device="cuda:0"); ws = torch.empty(E.ext_workspace_bytes(n), dtype=torch.uint8, device="cuda:0")
dv = lambda a: torch.from_numpy(a).to("cuda:0")
dR3, dPK3, dR1, dPK1 = dv(R3), dv(PK3), dv(R1), dv(PK1)
for rnd in range(2):
    print("host random z  best %.2f median %.2f ms" % best(lambda: E.verify_single_ext(h["u"], R3, PK3, h["m"])))
    print("host z = 1     best %.2f median %.2f ms" % best(lambda: E.verify_single_ext(h["u"], R1, PK1, h["m"])))
    print("dev  random z  %.2f ms" % dev(lambda: E.verify_single_ext_dev(b["u"], dR3, dPK3, b["m"], okd, ws)))
    print("dev  z = 1     %.2f ms" % dev(lambda: E.verify_single_ext_dev(b["u"], dR1, dPK1, b["m"], okd, ws)))
