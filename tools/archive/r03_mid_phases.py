#!/usr/bin/env python3
"""Round 3: phase stamps inside the two kernels of the mid-size route (workgroup 7 of each; 10 ns ticks).
Needs a full -DGS_EXP=5120 build (1024: the status-word reader, 4096: the stamps).
Usage: GPUSORT_LIB=gpusorting_amd/lib/libgpusort_exp5120f.so python tools/r03_mid_phases.py [log2 from=16] [to=22]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402
from gpusorting_amd import _lib  # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 22
lib = _lib.load()
fn = lib.gs_debug_read_status_words
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
n1 = ["entry", "own tile loaded", "ranked, counts published", "every row there", "bases", "staged", "scatter issued"]
n2 = ["entry", "bucket loaded", "pass 0 starts", "pass 1 starts", "pass 2 starts", "passes done", "stores issued"]
for lg in range(lo, hi + 1):
    n = 1 << lg
    k = torch.empty(n, dtype=torch.int32, device="cuda")
    s = g.OneSweep(n)
    rows = []
    for r in range(7):
        g.init_random(k, 10 + r, 0)
        torch.cuda.synchronize()
        s.sort(k)
        torch.cuda.synchronize()
        w = (C.c_uint32 * 32)()
        fn(s._h, w, None)
        t0 = w[16]
        a = [((w[16 + i] - t0) & 0xffffffff) / 100.0 for i in range(7)]
        b = [((w[24 + i] - t0) & 0xffffffff) / 100.0 for i in range(7)]
        rows.append((b[6], a, b))
    rows.sort(key=lambda x: x[0])
    _, a, b = rows[len(rows) // 2]
    print(f"2^{lg}: us after K1's entry (workgroup 7)")
    print("   K1: " + "  ".join(f"{nm} {v:.2f}" for nm, v in zip(n1, a)))
    print("   K2: " + "  ".join(f"{nm} {v:.2f}" for nm, v in zip(n2, b)))
    s.close()
