import os, sys, torch
sys.path.insert(0, os.getcwd())
import gpusorting_amd as g
for log2n in (22, 23, 24):
    n = 1 << log2n
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    for name, kw in (("default", {}), ("mid_path=0", dict(mid_path=0)), ("first_pass_big=0", dict(first_pass_big=0))):
        s = g.OneSweep(n, **kw)
        s.set_profiling(True)
        runs = []
        for it in range(12):
            g.init_random(dk, 10 + it, 0)
            s.sort(dk)
            p = s.get_profile()
            if it >= 2: runs.append(p)
        runs.sort(key=lambda r: r["total"])
        med = runs[len(runs)//2]
        print(f"2^{log2n} {name}: " + " ".join(f"{k}={v*1000:.1f}" for k, v in med.items()) + f" us -> {n/med['total']/1e6:.1f} GKeys/s", flush=True)
        s.close()
