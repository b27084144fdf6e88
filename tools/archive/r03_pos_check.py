#!/usr/bin/env python3
"""Round 3: the position-chain form of the sort (PF_POS) — exact results against torch.sort and per-kernel times, entropy presets 1..5.
Usage: GPUSORT_LIB=... python tools/r03_pos_check.py [log2n=28] [extra=0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = (1 << log2n) + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
vb = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 8: (u32, u64) pairs, value = original index
k = torch.empty(n, dtype=torch.int32, device="cuda")
v = torch.empty(n, dtype=torch.int64, device="cuda") if vb else None
for preset in range(5):
    s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
    s.set_profiling(True)
    best = None
    ok = True
    for r in range(4):
        g.init_random(k, 10 + r, preset)
        if vb:
            torch.arange(n, dtype=torch.int64, device="cuda", out=v)
        if r == 0:
            ref = torch.sort(k.to(torch.int64) & 0xFFFFFFFF, stable=True)
        s.sort(k, v)
        torch.cuda.synchronize()
        p = s.get_profile()
        if r == 0:
            ok = bool(((k.to(torch.int64) & 0xFFFFFFFF) == ref.values).all())
            if vb:
                ok = ok and bool((v == ref.indices).all())
            st = s.check_state()
            del ref
        elif best is None or p["total"] < best["total"]:
            best = p
    s.check()
    print(f"preset {preset + 1}: exact={ok} total={best['total']:.3f} ms = {n / best['total'] / 1e6:.1f} GKeys/s  hist={best['global_histogram']:.3f} "
          f"passes=[{best['pass0']:.3f} {best['pass1']:.3f} {best['pass2']:.3f} {best['pass3']:.3f}]  state={st}")
    s.close()
