#!/usr/bin/env python3
"""Why are passes 1-3 slower than pass 0?  Times stand-alone passes (position chains) on fresh data and on
data just written by a previous pass, next to the full sort's per-pass times; plus device-side totals at small n."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

n = 1 << 28
keys = torch.empty(n, dtype=torch.int32, device="cuda")
alt = torch.empty_like(keys)
s = g.OneSweep(n)
s.set_profiling(True)
acc = {}
for rep in range(6):
    g.init_random(keys, 10 + rep, 0)
    torch.cuda.synchronize()
    s.digit_pass(keys, alt, 0); a = s.get_profile()["pass0"]          # fresh input
    s.digit_pass(alt, keys, 1); b = s.get_profile()["pass0"]          # input written by the previous kernel
    s.digit_pass(keys, alt, 2); c = s.get_profile()["pass0"]
    torch.cuda.synchronize()
    # let the data age: 1 GiB of unrelated traffic, then the same kind of pass again
    junk = torch.empty(1 << 28, dtype=torch.int32, device="cuda"); junk.zero_(); junk2 = junk.clone(); torch.cuda.synchronize()
    s.digit_pass(alt, keys, 3); d = s.get_profile()["pass0"]
    g.init_random(keys, 10 + rep, 0); torch.cuda.synchronize()
    s.sort(keys, alt_keys=alt); p = s.get_profile()
    if rep:
        for k, v in (("standalone pass0 fresh", a), ("standalone pass1 on just-written", b), ("standalone pass2 on just-written", c),
                     ("standalone pass3 on aged data", d), ("sort pass0", p["pass0"]), ("sort pass1", p["pass1"]),
                     ("sort pass2", p["pass2"]), ("sort pass3", p["pass3"])):
            acc[k] = acc.get(k, 0) + v / 5
for k, v in acc.items():
    print(f"{k:36s} {v:.3f} ms")
s.close()
print("--- device-side total per sort at small n (HIP events around the whole sort)")
for lg in (10, 14, 16, 18, 20, 22):
    m = 1 << lg
    k = torch.empty(m, dtype=torch.int32, device="cuda"); al = torch.empty_like(k)
    ss = g.OneSweep(m); ss.set_profiling(True)
    tot = 0
    for rep in range(6):
        g.init_random(k, rep + 1, 0); torch.cuda.synchronize()
        ss.sort(k, alt_keys=al); p = ss.get_profile()
        if rep: tot += p["total"] / 5
    print(f"2^{lg}: {tot*1e3:.1f} us  (clear {p['clear']*1e3:.1f} hist {p['global_histogram']*1e3:.1f} scan {p['scan']*1e3:.1f} passes {[round(p[f'pass{i}']*1e3,1) for i in range(4)]})")
    ss.close()
