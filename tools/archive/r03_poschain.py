#!/usr/bin/env python3
"""Round 3: do POSITION chains keep a skewed sort's passes 1-3 at pass-0 speed?  Four stand-alone passes (GlobalHistogram of one
digit + Scan + one DigitBinningPass over 16 position segments) chained like a sort: pass p reads what pass p-1 wrote.
Usage: python tools/r03_poschain.py [log2n=28] [reps=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << log2n
a = torch.empty(n, dtype=torch.int32, device="cuda")
b = torch.empty(n, dtype=torch.int32, device="cuda")
s = g.OneSweep(n)
s.set_profiling(True)
for preset in range(5):
    best = [None] * 4
    besth = [None] * 4
    for r in range(reps):
        g.init_random(a, 10 + r, preset)
        x, y = a, b
        for p in range(4):
            s.digit_pass(x, y, p)
            torch.cuda.synchronize()
            prof = s.get_profile()
            if best[p] is None or prof["pass0"] < best[p]:
                best[p] = prof["pass0"]
            if besth[p] is None or prof["global_histogram"] < besth[p]:
                besth[p] = prof["global_histogram"]
            x, y = y, x
        ok = g.validate(x) == 0
    s2 = g.OneSweep(n)
    s2.set_profiling(True)
    g.init_random(a, 10, preset)
    s2.sort(a)
    torch.cuda.synchronize()
    q = s2.get_profile()
    print(f"preset {preset + 1}: position-chain passes [{best[0]:.3f} {best[1]:.3f} {best[2]:.3f} {best[3]:.3f}] one-digit histograms "
          f"[{besth[0]:.3f} {besth[1]:.3f} {besth[2]:.3f} {besth[3]:.3f}] sorted={ok} | sort: hist {q['global_histogram']:.3f} passes "
          f"[{q['pass0']:.3f} {q['pass1']:.3f} {q['pass2']:.3f} {q['pass3']:.3f}] total {q['total']:.3f}")
    s2.close()
