#!/usr/bin/env python3
"""torch.sort (rocPRIM underneath) vs gpusorting_amd.sort_ on the same tensors.  Usage: torch_compare.py [log2=28]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << lg


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        a.record(); fn(); b.record(); b.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


for dtype in (torch.int32, torch.float32):
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    g.init_random(src, 10, 0)
    src = src.view(dtype)
    work = src.clone()
    t_torch = timed(lambda: torch.sort(src))
    def ours():
        work.copy_(src)          # sort_ is in place: the copy is inside the timing, torch.sort's output allocation too
        g.sort_(work)
    t_ours = timed(ours)
    ref = torch.sort(src).values
    if dtype == torch.int32:
        ok = bool((work == ref).all().item())
    else:  # torch orders NaNs last and -0 == +0; compare where the orders are defined alike
        ok = bool((work[~torch.isnan(work)] == ref[~torch.isnan(ref)]).all().item())
    t_copy = timed(lambda: work.copy_(src))
    print(f"2^{lg} {str(dtype):14s} torch.sort {t_torch:7.3f} ms   gpusorting_amd.sort_ (+copy) {t_ours:7.3f} ms "
          f"(copy alone {t_copy:.3f})   x{t_torch / t_ours:.2f}   equal={ok}")
