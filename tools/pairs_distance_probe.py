"""How far apart should a key buffer and its value buffer be?  (u32, u32) / (u32, u64) pairs at 2^28, default routing; ONE arena:
keys at 0, values at D, alt keys at A, alt values at A + D.  Per-slot times against D.   python tools/pairs_distance_probe.py [vb=4]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

n = 1 << 28
vb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
vw = vb // 4
vdt = torch.int32 if vb == 4 else torch.int64
GB = 1 << 28  # int32 words per GiB
arena = torch.empty(13 * GB, dtype=torch.int32, device="cuda")
base = (-(arena.data_ptr() // 4)) % GB   # start the layout on a GiB boundary
for A_gb, D_q in ((6, 4), (6, 5), (6, 6), (6, 7), (6, 8), (6, 10), (6, 12), (6.5, 4), (6.5, 8), (5, 4), (5, 8)):
    D = D_q * GB // 4 if vb == 4 else max(D_q, 4) * GB // 4
    A = int(A_gb * GB)
    if D < n or D + n * vw > A or A + D + n * vw > arena.numel() - base:
        continue
    k = arena[base:base + n]
    v = arena[base + D:base + D + n * vw].view(vdt)
    ak = arena[base + A:base + A + n]
    av = arena[base + A + D:base + A + D + n * vw].view(vdt)
    s = g.OneSweep(n, mode=g.MODE_PAIRS, value_bytes=vb)
    s.set_profiling(True)
    runs = []
    for it in range(6):
        g.init_random(k, 10 + it, 0, v)
        s.sort(k, v, alt_keys=ak, alt_values=av)
        if it:
            runs.append(s.get_profile())
    runs.sort(key=lambda r: r["total"])
    m = runs[len(runs) // 2]
    print(f"values at keys + {D_q / 4:.2f} GiB, alt at keys + {A_gb} GiB: pass0={m['pass0']:.4f} pass1={m['pass1']:.4f} pass2={m['pass2']:.4f} total={m['total']:.4f} -> {n / m['total'] / 1e6:.1f} GKeys/s", flush=True)
    s.close()
