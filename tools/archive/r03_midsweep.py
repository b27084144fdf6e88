#!/usr/bin/env python3
"""Round 3: keys-only sorts of 2^21 .. 2^27 keys, back-to-back (20 per point), under the environment's routing.
Usage: [GPUSORT_POS=2 GPUSORT_POS_MIN_LOG2=22] python tools/r03_midsweep.py [vb=0] [log2 from=21] [log2 to=27]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

vb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
row = []
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 21
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 27
for lg in range(lo, hi + 1):
    n = 1 << lg
    nb = max(2, min(20, (1 << 30) // (n * 4)))
    ks = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nb)]
    vs = [torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda") for _ in range(nb)] if vb else [None] * nb
    alt = torch.empty(n, dtype=torch.int32, device="cuda")
    valt = torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda") if vb else None
    s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(4):
        for i in range(nb):
            g.init_random(ks[i], 10 + i + 100 * rep, 0, vs[i])
        torch.cuda.synchronize()
        a.record()
        for i in range(nb):
            s.sort(ks[i], vs[i], alt_keys=alt, alt_values=valt)
        b.record()
        b.synchronize()
        if rep:
            best = min(best, a.elapsed_time(b) / nb * 1e3)
    s.check()
    ok = g.validate(ks[-1], vs[-1] if vb == 4 else None) == 0
    s.set_profiling(True)
    g.init_random(ks[0], 5, 0, vs[0])
    s.sort(ks[0], vs[0], alt_keys=alt, alt_values=valt)
    torch.cuda.synchronize()
    p = s.get_profile()
    s.close()
    print(f"2^{lg}: {best:8.1f} us  {n / best / 1e3:7.2f} GKeys/s  sorted={ok}  profiled: hist {p['global_histogram']*1e3:.1f} scan {p['scan']*1e3:.1f} "
          f"passes {p['pass0']*1e3:.1f} {p['pass1']*1e3:.1f} {p['pass2']*1e3:.1f} {p['pass3']*1e3:.1f}", flush=True)
