"""Does splitting one batch over two streams hide the fill/drain phases of the two kernels?"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from schnorr_amd import engine as E, workload as W
E.init(0)
dev = "cuda:0"
n = 1 << 20
b = W.gen_single(n, 2321, tamper=False)
ok = torch.empty(n, dtype=torch.uint8, device=dev)
streams = [torch.cuda.Stream() for _ in range(4)]
NS = 2

def run(parts):
    sz = n // parts
    wss = [torch.empty(E.workspace_bytes(sz), dtype=torch.uint8, device=dev) for _ in range(min(parts, NS))]
    def step():
        for p in range(parts):
            s = streams[p % NS] if parts > 1 else torch.cuda.current_stream()
            sl = slice(p * sz, (p + 1) * sz)
            E.verify_single_dev(b["u"][sl], b["R"][sl], b["PK"][sl], b["m"][sl], ok[sl], wss[p % len(wss)], stream=s)
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    assert bool(ok.all())
    return dt * 1e3
for NS in (2, 3, 4):
    for parts in (1, 8, 16, 32, 64, 16):
        t = run(parts)
        print("streams %d parts %2d: %.3f ms per 2^20 -> %.2f M/s" % (NS, parts, t, n / t / 1e3))
