#!/usr/bin/env python3
"""One fast-accept workload, repeated (for profilers):

    python tools/rlc_case.py [log2n=20] [scheme=single] [valid|one|graded] [history=0] [reps=4] [window_bits=0]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from schnorr_amd import engine as E  # noqa: E402
from schnorr_amd import workload as W  # noqa: E402

E.init(0)
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
scheme = sys.argv[2] if len(sys.argv) > 2 else "single"
case = sys.argv[3] if len(sys.argv) > 3 else "valid"
history = int(sys.argv[4]) if len(sys.argv) > 4 else 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
bits = int(sys.argv[6]) if len(sys.argv) > 6 else 0
COLS = {"single": ("u", "R", "PK", "m"), "double": ("u", "R", "Rp", "PK", "PKp", "m"), "vargen": ("u", "R", "PK", "Gen", "m")}[scheme]
b = getattr(W, "gen_" + scheme)(n, seed=2321, tamper=case == "graded")
if case == "one":
    b["u"][(5 * n) // 8 + 77, 3] ^= 0x10
    b["expected"][(5 * n) // 8 + 77] = 0
ws = torch.empty(E.rlc_workspace_bytes(n, bits), dtype=torch.uint8, device="cuda:0")
ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
word = torch.zeros(1, dtype=torch.int32).pin_memory()
for _ in range(reps):
    E.rlc_history(0, history)
    getattr(E, "verify_%s_rlc_dev" % scheme)(*[b[k] for k in COLS], ok, ws, window_bits=bits, accepted_out=word)
    torch.cuda.synchronize()
    assert torch.equal(ok, b["expected"])
print("accepted", int(word[0]))
