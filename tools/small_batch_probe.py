import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from schnorr_amd import engine as E, workload as W
E.init(0)
n=1024
b=W.gen_single(n, seed=5)
h={k:b[k].cpu().numpy() for k in ("u","R","PK","m")}
for _ in range(20): E.verify_single(h["u"],h["R"],h["PK"],h["m"])
t0=time.perf_counter()
for _ in range(200): E.verify_single(h["u"],h["R"],h["PK"],h["m"])
print("host call %.3f ms" % ((time.perf_counter()-t0)/200*1e3))
ok=torch.empty(n,dtype=torch.uint8,device="cuda:0"); ws=torch.empty(E.workspace_bytes(n),dtype=torch.uint8,device="cuda:0")
for _ in range(20): E.verify_single_dev(b["u"],b["R"],b["PK"],b["m"],ok,ws)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(200): E.verify_single_dev(b["u"],b["R"],b["PK"],b["m"],ok,ws)
torch.cuda.synchronize()
print("device call (back to back) %.3f ms" % ((time.perf_counter()-t0)/200*1e3))
