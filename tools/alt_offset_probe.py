import os, sys, torch
sys.path.insert(0, os.getcwd())
import gpusorting_amd as g
n = 1 << 28
big = torch.empty(n + (8 << 20), dtype=torch.int32, device="cuda")
dk = torch.empty(n, dtype=torch.int32, device="cuda")
for off_kb in (0, 4, 68, 260, 1028, 2052, 4100, 16388):
    off = off_kb * 256   # int32 elements
    alt = big[off:off + n]
    s = g.OneSweep(n)
    s.set_profiling(True)
    runs = []
    for it in range(7):
        g.init_random(dk, 10 + it, 0)
        s.sort(dk, alt_keys=alt)
        if it: runs.append(s.get_profile())
    runs.sort(key=lambda r: r["total"])
    m = runs[len(runs) // 2]
    print(f"alt offset {off_kb:6d} KiB (keys {dk.data_ptr():#x} alt {alt.data_ptr():#x}): " + " ".join(f"{k}={v:.4f}" for k, v in m.items()), flush=True)
    s.close()
