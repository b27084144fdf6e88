#!/usr/bin/env python3
"""Achievable HBM rates on this box (2^28 uint32 = 1 GiB buffers): plain 16-byte copies, read-only sweep,
and the DigitBinningPass-shaped tile copy."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpusorting_amd import _lib  # noqa: E402

n = 1 << 28
lib = _lib.load_tuning()  # calibration kernels live in the tuning build
a = torch.empty(n, dtype=torch.int32, device="cuda"); a.random_()
b = torch.empty_like(a)
sp = int(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


for name, t, k, byts in (("copy 16B default policy", 0, 0, 8), ("copy 16B nt loads", 0, 1, 8), ("copy 16B nt loads+stores", 0, 2, 8),
                         ("read-only 16B sweep", 0, 3, 4), ("tile-shaped copy 512x32? (512x16)", 512, 16, 8),
                         ("tile-shaped copy 1024x16", 1024, 16, 8), ("tile-shaped copy 256x32", 256, 32, 8)):
    ms = timed(lambda: lib.gs_debug_copy_floor(a.data_ptr(), b.data_ptr(), n, t, k, sp))
    print(f"{name:36s} {ms:.3f} ms  {byts*n/ms/1e6:7.0f} GB/s")
ms = timed(lambda: b.copy_(a))
print(f"{'torch copy_':36s} {ms:.3f} ms  {8*n/ms/1e6:7.0f} GB/s")
