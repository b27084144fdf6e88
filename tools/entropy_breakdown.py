#!/usr/bin/env python3
"""Per-kernel HIP-event breakdown of the sort for every entropy preset (keys / u32 values / u64 values).
Usage: entropy_breakdown.py [log2_keys=28] [reps=5] [value_bytes=0,8]   (library: env GPUSORT_LIB)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402
from gpusorting_amd import _lib  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
vbs = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,8").split(",")]
n = 1 << lg
print(f"lib={_lib.LIB_PATH} n=2^{lg}")
for vb in vbs:
    k = torch.empty(n, dtype=torch.int32, device="cuda")
    v = None if not vb else torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
    s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
    s.set_profiling(True)
    for preset in range(5):
        best = None
        for r in range(reps):
            g.init_random(k, 10 + r, preset, v)
            s.sort(k, v)
            torch.cuda.synchronize()
            p = s.get_profile()
            best = p if best is None or p["total"] < best["total"] else best
        ok = bool((k[1:].to(torch.int64) & 0xffffffff >= k[:-1].to(torch.int64) & 0xffffffff).all().item()) if n <= (1 << 26) else "-"
        print(f"vb={vb} preset {preset + 1}: {n / best['total'] / 1e6:7.2f} GKeys/s total={best['total']:.3f} ms  hist={best['global_histogram']:.3f} "
              f"scan={best['scan']:.3f} passes=[{best['pass0']:.3f} {best['pass1']:.3f} {best['pass2']:.3f} {best['pass3']:.3f}] sorted={ok}",
              flush=True)
    s.close()
