#!/usr/bin/env python3
"""REPS sorts of 2^LOG keys (entropy preset index P, value bytes VB) for profilers.
Usage: sort_loop.py [log2=28] [reps=3] [preset=0] [vb=0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
preset = int(sys.argv[3]) if len(sys.argv) > 3 else 0
vb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = 1 << lg
k = torch.empty(n, dtype=torch.int32, device="cuda")
v = None if not vb else torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
for r in range(reps):
    g.init_random(k, 10 + r, preset, v)
    s.sort(k, v)
torch.cuda.synchronize()
s.check()
print("done")
