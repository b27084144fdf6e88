#!/usr/bin/env python3
"""Tile shapes against each other on one box, per entropy preset: best-of-N whole-sort time and pass times.
Usage: shape_ab.py vb preset shape [shape ...]      (shape = TxK or 'auto')"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

vb, preset = int(sys.argv[1]), int(sys.argv[2])
shapes = sys.argv[3:]
n = 1 << 28
k = torch.empty(n, dtype=torch.int32, device="cuda")
v = None if not vb else torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
res = {}
for rnd in range(3):
    for sh in shapes:
        s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
        if sh != "auto":
            t, kk = (int(x) for x in sh.split("x"))
            s.set_shape(t, kk)
        s.set_profiling(True)
        for r in range(4):
            g.init_random(k, 10 + r, preset, v)
            s.sort(k, v)
            torch.cuda.synchronize()
            p = s.get_profile()
            if r and (sh not in res or p["total"] < res[sh]["total"]):
                res[sh] = p
        s.close()
for sh in shapes:
    p = res[sh]
    print(f"vb={vb} preset={preset + 1} shape={sh:8s} total={p['total']:.3f} ms passes=[{p['pass0']:.3f} {p['pass1']:.3f} {p['pass2']:.3f} {p['pass3']:.3f}] "
          f"hist={p['global_histogram']:.3f}")
