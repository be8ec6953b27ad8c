import time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from schnorr_amd import engine as E
t0=time.perf_counter(); E.init(0); torch.cuda.synchronize(); t1=time.perf_counter()
E.shutdown(); t2=time.perf_counter(); E.init(0); torch.cuda.synchronize(); t3=time.perf_counter()
print("init %.1f ms, re-init %.1f ms" % ((t1-t0)*1e3, (t3-t2)*1e3))
