#!/usr/bin/env python3
"""Histogram kernel time with 1 pass (np=1, one LDS atomic per key) vs 4 passes (np=4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g
n = 1 << 28
keys = torch.empty(n, dtype=torch.int32, device="cuda"); alt = torch.empty_like(keys)
s = g.OneSweep(n); s.set_profiling(True)
a = b = 0.0
for rep in range(6):
    g.init_random(keys, 10 + rep, 0); torch.cuda.synchronize()
    s.digit_pass(keys, alt, 0); p1 = s.get_profile()
    g.init_random(keys, 10 + rep, 0); torch.cuda.synchronize()
    s.sort(keys, alt_keys=alt); p4 = s.get_profile()
    if rep: a += p1["global_histogram"] / 5; b += p4["global_histogram"] / 5
print(f"global_histogram np=1: {a:.3f} ms   np=4: {b:.3f} ms   (read-only sweep floor 0.166 ms)")
