#!/usr/bin/env python3
"""Timeline of ONE host-buffer call (affine | ext | wire | e2e-single | e2e-double | e2e-vargen |
e2e-*-streamed) from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 tools/host_timeline.py run wire
    python3 tools/host_timeline.py report DIR [rows]

The e2e kinds run verify_batch* over typed objects (tools/libvb_e2e.so); `-streamed`: four batches, two
in flight, as the measured call.  `report DIR rows` also lists every dispatch and copy of the measured
call with its hardware queue and stream.

`run` performs two warm calls, prints a wall-clock marker, then the measured call (DSV_PIPE_TRACE=1
in the environment adds the pipeline's own per-call line on stderr).  `report` reads the kernel
trace, takes the dispatches of the LAST call, and prints per kernel family: launches, summed
duration, and for the whole call: first dispatch -> last completion, time with at least one kernel
running, time with none (the GPU waiting for the host or for a copy), and the idle gaps > 50 us with
the kernels around them."""
import csv
import glob
import os
import sys
import time


def run(kind):
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import torch  # noqa: F401
    from schnorr_amd import engine as E
    from schnorr_amd import workload as W
    E.init(0)
    n = 1 << int(os.environ.get("LOG2N", "20"))
    if not kind.startswith("e2e"):
        b = W.gen_single(n, seed=2321)
        h = {k: b[k].cpu().numpy() for k in ("u", "R", "PK", "m")}
        want = b["expected"].cpu().numpy()
    if kind.startswith("e2e"):
        import ctypes
        L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvb_e2e.so"))
        scheme = kind.split("-")[1]
        streamed = kind.endswith("streamed")
        if scheme == "vargen":
            n = 1 << int(os.environ.get("LOG2N", "18"))
        gen = {"single": W.gen_single, "double": W.gen_double, "vargen": W.gen_vargen}[scheme]
        keys = {"single": ("u", "R", "PK", "m"), "double": ("u", "R", "Rp", "PK", "PKp", "m"),
                "vargen": ("u", "R", "PK", "Gen", "m")}[scheme]
        k = ("single", "double", "vargen").index(scheme)
        b = gen(n, seed=2321)
        want = b["expected"].cpu().numpy()
        hh = [b[x].cpu().numpy() for x in keys]
        p = lambda a: ctypes.c_void_p(a.ctypes.data)
        prep = (L.vb_e2e_prepare, L.vb_e2e_prepare_double, L.vb_e2e_prepare_vargen)[k]
        runf = (L.vb_e2e_run, L.vb_e2e_run_double, L.vb_e2e_run_vargen)[k]
        prep(*([p(a) for a in hh] + [ctypes.c_size_t(n), ctypes.c_int(8)]))
        ok = np.zeros(n, dtype=np.uint8)
        ms = ctypes.c_double(0)

        def fn():
            if streamed:
                assert L.vb_e2e_run_streamed(ctypes.c_int(k), ctypes.c_int(4), ctypes.c_int(2), p(ok), ctypes.byref(ms)) == 0
            else:
                assert runf(p(ok), ctypes.byref(ms)) == 0
            return ok
    elif kind == "affine":
        fn = lambda: E.verify_single(h["u"], h["R"], h["PK"], h["m"])
    elif kind == "wire":
        sig = np.ascontiguousarray(np.concatenate([h["u"], E.compress_points(h["R"])], axis=1))
        pk = E.compress_points(h["PK"])
        fn = lambda: E.verify_single_wire(sig, pk, h["m"])
    else:
        z = np.random.default_rng(1).integers(0, 256, (n, 32), dtype=np.uint8)
        z[:, 31] = 0
        z[:, 0] |= 1
        proj = lambda a: np.concatenate([E.debug_fq_mul(np.ascontiguousarray(a[:, :32]), z),
                                         E.debug_fq_mul(np.ascontiguousarray(a[:, 32:]), z), z], axis=1)
        R3, PK3 = proj(h["R"]), proj(h["PK"])
        if kind == "extz1":  # every point with z = 1 (affine-lifted input): the inversions are 1/1
            one = np.zeros((n, 32), np.uint8)
            one[:, 0] = 1
            R3 = np.ascontiguousarray(np.concatenate([h["R"], one], axis=1))
            PK3 = np.ascontiguousarray(np.concatenate([h["PK"], one], axis=1))
        fn = lambda: E.verify_single_ext(h["u"], R3, PK3, h["m"])
    fn()
    fn()
    walls = []
    for _ in range(int(os.environ.get("PRE_CALLS", "0"))):   # wall times of untraced-looking calls (bimodality check)
        t0 = time.perf_counter()
        fn()
        walls.append((time.perf_counter() - t0) * 1e3)
    if walls:
        print("pre-calls ms: " + " ".join("%.2f" % w for w in walls))
    torch.cuda.synchronize()
    time.sleep(0.05)                       # a visible gap in the trace in front of the measured call
    t0 = time.perf_counter()
    ok = fn()
    dt = time.perf_counter() - t0
    assert (ok == want).all()
    print("%s host call, n = 2^%d: %.2f ms wall = %.2f M/s" % (kind, n.bit_length() - 1, dt * 1e3, n / dt / 1e6))


def report(d, list_rows=False):
    rows, copies, meta = [], [], {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
            rows.append(key)
            meta[key] = (r.get("Queue_Id", "-"), r.get("Stream_Id", "-"))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?"),
                           r.get("Stream_Id", "-")))
    rows.sort()
    # the measured call = everything after the last gap of > 30 ms between dispatches
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - max(x[1] for x in rows[max(0, i - 40):i]) > 30e6:
            cut = i
    call = rows[cut:]
    t0, t1 = call[0][0], max(r[1] for r in call)
    fam = {}
    for s, e, k in call:
        k = k.split("(")[0].replace("dsv::", "").replace("void ", "")[:40]
        a = fam.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    print("dispatches %d, first start -> last end %.3f ms" % (len(call), (t1 - t0) / 1e6))
    for k, (c, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("  %-42s x%-4d %8.3f ms summed" % (k, c, ms))
    ev = sorted([(s, 1, k) for s, e, k in call] + [(e, -1, k) for s, e, k in call])
    depth, busy, last, idle_gaps, prev_name = 0, 0.0, t0, [], ""
    for t, dlt, k in ev:
        if depth > 0:
            busy += t - last
        elif t > last and t - last > 50e3:
            idle_gaps.append(((last - t0) / 1e6, (t - last) / 1e6, prev_name[:30], k[:30]))
        last = t
        depth += dlt
        prev_name = k
    print("GPU running a kernel %.3f ms, idle %.3f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
    for at, ln, a, b in idle_gaps:
        print("  idle %.3f ms at +%.3f ms  (after %s, before %s)" % (ln, at, a, b))
    if list_rows:
        allr = [(s_, e_, k.split("(")[0].replace("dsv::", "").replace("void ", "")[:34], meta[(s_, e_, k)][0], meta[(s_, e_, k)][1])
                for s_, e_, k in call]
        allr += [(s_, e_, k, "-", st) for s_, e_, k, st in copies if t0 - 2e6 <= s_ <= t1]
        for s_, e_, k, q, st in sorted(allr):
            print("%8.3f %8.3f %6.3f  %-34s q=%s st=%s" % ((s_ - t0) / 1e6, (e_ - t0) / 1e6, (e_ - s_) / 1e6, k, q, st))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        report(sys.argv[2], len(sys.argv) > 3 and sys.argv[3] == "rows")
