#!/usr/bin/env python3
"""Which tile shape wins at which size on the general (six-launch) path: back-to-back sorts per (value bytes, shape, size).
Usage: shape_by_size.py [vb=0] [sizes=22,23,24,25,26] [shapes=512x32,512x16,1024x16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

vb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "22,23,24,25,26").split(",")]
shapes = (sys.argv[3] if len(sys.argv) > 3 else "512x32,512x16,1024x16").split(",")
vdt = torch.int32 if vb == 4 else torch.int64
for lg in sizes:
    n = 1 << lg
    nb = max(2, min(16, (1 << 28) // n))
    keys = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nb)]
    vals = [torch.empty(n, dtype=vdt, device="cuda") for _ in range(nb)] if vb else [None] * nb
    alt = torch.empty(n, dtype=torch.int32, device="cuda")
    valt = torch.empty(n, dtype=vdt, device="cuda") if vb else None
    row = []
    for shape in shapes:
        t, k = (int(x) for x in shape.split("x"))
        s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
        s.set_mid_path(False)
        s.set_shape(t, k)
        best = 1e9
        for rep in range(4):
            for i in range(nb):
                g.init_random(keys[i], 10 + i + rep, 0, vals[i])
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(nb):
                s.sort(keys[i], vals[i], alt_keys=alt, alt_values=valt)
            b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b) / nb * 1e3)
        row.append(f"{shape} {best:7.1f} us")
        s.close()
    print(f"vb={vb} 2^{lg}: " + "   ".join(row), flush=True)
