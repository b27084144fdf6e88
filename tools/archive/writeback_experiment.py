#!/usr/bin/env python3
"""Does a kernel pay for the dirty lines its predecessor left in the memory-side cache?  A read-only sweep, then four
tile-shaped copies a->b, b->a, ..., each timed with its own HIP events (2^28 uint32)."""
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
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
acc = [0.0] * 5
reps = 6
for r in range(reps):
    torch.cuda.synchronize()
    ev[0].record()
    lib.gs_debug_copy_floor(a.data_ptr(), b.data_ptr(), n, 0, 3, sp)  # read-only sweep of a
    ev[1].record()
    for i in range(4):
        src, dst = (a, b) if i % 2 == 0 else (b, a)
        lib.gs_debug_copy_floor(src.data_ptr(), dst.data_ptr(), n, 256, 32, sp)
        ev[2 + i].record()
    torch.cuda.synchronize()
    if r:
        for i in range(5):
            acc[i] += ev[i].elapsed_time(ev[i + 1]) / (reps - 1)
print(f"read-only sweep            {acc[0]:.3f} ms")
for i in range(4):
    print(f"tile copy #{i + 1} after {'the sweep' if i == 0 else 'a copy   '}  {acc[1 + i]:.3f} ms")
