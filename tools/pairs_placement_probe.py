"""(u32, u32) pairs at 2^28 on the default routing: keys | values | alt keys | alt values carved out of ONE arena with a gap of g KiB
in front of each of the last three — and, second block, four separate torch allocations in different orders.  Per-slot times.
  python tools/pairs_placement_probe.py [value_bytes=4]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

n = 1 << 28
vb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
vw = vb // 4


def run(tag, k, v, ak, av):
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
    print(f"{tag:34s} pass0={m['pass0']:.4f} pass1={m['pass1']:.4f} pass2={m['pass2']:.4f} total={m['total']:.4f}  "
          f"k={k.data_ptr():#x} v={v.data_ptr():#x} ak={ak.data_ptr():#x} av={av.data_ptr():#x}", flush=True)
    s.close()


vdt = torch.int32 if vb == 4 else torch.int64
arena = torch.empty(n * (2 + 2 * vw) + (64 << 20), dtype=torch.int32, device="cuda")
for gap_kb in (0, 4, 36, 68, 260, 1028, 4100):
    gw = gap_kb * 256
    o = 0
    k = arena[o:o + n]; o += n + gw
    v = arena[o:o + n * vw].view(vdt); o += n * vw + gw
    ak = arena[o:o + n]; o += n + gw
    av = arena[o:o + n * vw].view(vdt)
    run(f"one arena, gaps {gap_kb} KiB", k, v, ak, av)
del arena
torch.cuda.empty_cache()
for order in ("k v ak av", "ak av k v", "k ak v av", "v k av ak"):
    t = {}
    for name in order.split():
        t[name] = torch.empty(n if name in ("k", "ak") else n, dtype=torch.int32 if name in ("k", "ak") else vdt, device="cuda")
    run(f"separate allocations, order {order}", t["k"], t["v"], t["ak"], t["av"])
    del t
    torch.cuda.empty_cache()
