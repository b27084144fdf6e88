#!/usr/bin/env python3
"""Round 3: 64-bit keys, one GlobalHistogram + Scan for eight passes against one per word (GPUSORT_KEY64_SWEEPS=2), same process.
Usage: python tools/r03_keys64.py [log2n=27] [reps=5]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << lg
src = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda")
k = torch.empty_like(src)
for kind, mask in (("uniform", -1), ("values below 2^40", (1 << 40) - 1), ("values below 2^32", (1 << 32) - 1)):
    for sweeps in ("2", "1", "2", "1"):
        os.environ["GPUSORT_KEY64_SWEEPS"] = sweeps
        s = g.OneSweep(n, key_type=g.KEY_UINT64)
        s.set_profiling(True)
        best = None
        for r in range(reps):
            k.copy_(src & mask if mask != -1 else src)
            torch.cuda.synchronize()
            s.sort(k)
            torch.cuda.synchronize()
            p = s.get_profile()
            if r and (best is None or p["total"] < best["total"]):
                best = p
        ok = g.validate(k, key_type=g.KEY_UINT64) == 0
        s.close()
        print(f"2^{lg} u64 keys, {kind:18s} sweeps={sweeps}: {best['total']:.3f} ms = {n / best['total'] / 1e6:6.2f} GKeys/s  hist {best['global_histogram']:.3f} scan {best['scan']:.3f} "
              f"pass0..2 {best['pass0']:.3f} {best['pass1']:.3f} {best['pass2']:.3f} rest {best['pass3']:.3f}  sorted={ok}", flush=True)
