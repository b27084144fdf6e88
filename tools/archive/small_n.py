#!/usr/bin/env python3
"""Sorts 2^LOG keys REPS times back to back (for rocprofv3 --kernel-trace at small n). Usage: small_n.py [log2=16] [reps=50]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n = 1 << lg
keys = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(reps)]
alt = torch.empty(n, dtype=torch.int32, device="cuda")
for i, k in enumerate(keys):
    g.init_random(k, 10 + i, 0)
s = g.OneSweep(n)
s.sort(keys[0], alt_keys=alt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in keys:
    s.sort(k, alt_keys=alt)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"2^{lg}: {dt/reps*1e6:.1f} us per sort (host wall, {reps} back-to-back)")
