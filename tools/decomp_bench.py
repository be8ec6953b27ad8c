import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from schnorr_amd import engine as E, workload as W
E.init(0)
n = 1 << 18
b = W.gen_single(n, 5, tamper=False)
comp = b["PK"][:, 32:].clone().contiguous()
comp[:, 31] |= (b["PK"][:, 0] & 1) << 7
out = torch.empty((n, 64), dtype=torch.uint8, device="cuda:0")
ok = torch.empty(n, dtype=torch.uint8, device="cuda:0")
E.decompress_points_dev(comp, out, ok)
torch.cuda.synchronize()
assert torch.equal(out, b["PK"]) and bool(ok.all())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    E.decompress_points_dev(comp, out, ok)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("decompress: %.3f ms for 2^18 points -> %.1f M points/s" % (ms, n / ms / 1e3))
