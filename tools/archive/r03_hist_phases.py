#!/usr/bin/env python3
"""Round 3: where the GlobalHistogram kernel spends its time at mid sizes — phase stamps (10 ns ticks) of its first and last
workgroup.  Needs a -DGS_MINIMAL -DGS_EXP=5120 build (1024: the status-word reader, 4096: the stamps).
Usage: GPUSORT_LIB=gpusorting_amd/lib/libgpusort_exp5120.so GPUSORT_MID_PATH=0 python tools/r03_hist_phases.py [log2 from=22] [to=25]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402
from gpusorting_amd import _lib  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 22
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 25
lib = _lib.load()
fn = lib.gs_debug_read_status_words
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
names = ["entry", "LDS zeroed", "1st work item counted", "probe done", "loop end", "folded", "slice stored", "stores drained"]
print("# stamps in us after the first workgroup's entry; first workgroup | last workgroup;  event time of the histogram + reduce kernels")
for lg in range(lo, hi + 1):
    n = 1 << lg
    k = torch.empty(n, dtype=torch.int32, device="cuda")
    s = g.OneSweep(n)
    s.set_profiling(True)
    rows = []
    for r in range(6):
        g.init_random(k, 10 + r, 0)
        torch.cuda.synchronize()
        s.sort(k)
        torch.cuda.synchronize()
        w = (C.c_uint32 * 32)()
        fn(s._h, w, None)
        t0 = w[16]
        a = [((w[16 + i] - t0) & 0xffffffff) / 100.0 for i in range(8)]
        b = [((w[24 + i] - t0) & 0xffffffff) / 100.0 if w[24 + i] else float("nan") for i in range(8)]
        rows.append((s.get_profile()["global_histogram"] * 1e3, a, b))
    rows.sort(key=lambda x: x[0])
    ev, a, b = rows[len(rows) // 2]
    print(f"2^{lg}: events {ev:.1f} us")
    for i in range(8):
        print(f"    {names[i]:24s} {a[i]:7.2f} | {b[i]:7.2f}")
    s.close()
