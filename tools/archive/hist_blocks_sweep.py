#!/usr/bin/env python3
"""GlobalHistogram time against the number of workgroups at mid sizes (each workgroup closes with one global atomic per
non-empty bin of its 4 x 4096-bin LDS histograms).  Run once per GPUSORT_HIST_BLOCKS value; prints hist/total us per size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

for lg in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "21,22,23,24,25,26".split(","))]:
    n = 1 << lg
    k = torch.empty(n, dtype=torch.int32, device="cuda")
    s = g.OneSweep(n)
    s.set_profiling(True)
    best = None
    for r in range(8):
        g.init_random(k, 3 + r, 0)
        s.sort(k)
        torch.cuda.synchronize()
        p = s.get_profile()
        if r and (best is None or p["global_histogram"] < best["global_histogram"]):
            best = p
    print(f"blocks={os.environ.get('GPUSORT_HIST_BLOCKS', 'default'):>7s} 2^{lg}: hist {best['global_histogram'] * 1e3:6.1f} us  total {best['total'] * 1e3:7.1f} us")
    s.close()
