import os, sys, torch
sys.path.insert(0, os.getcwd())
import gpusorting_amd as g
for log2n in (23, 24, 25, 26):
    n = 1 << log2n
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    for name, kw in (("default-lsd", dict(plan=1)), ("two-level", dict(plan=2, position_chains_min_log2=20))):
        s = g.OneSweep(n, **kw)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for it in range(12):
            g.init_random(dk, 10 + it, 0)
            ev0.record(); s.sort(dk); ev1.record(); torch.cuda.synchronize()
            if it >= 2: ts.append(ev0.elapsed_time(ev1))
        ts.sort()
        print(f"2^{log2n} {name}: median {ts[len(ts)//2]*1000:.1f} us -> {n/ts[len(ts)//2]/1e6:.1f} GKeys/s {s.last_plan()}", flush=True)
        s.close()
