"""Where the two-level plan starts to pay: default routing with plan 1 (LSD passes) against the forced two-level plan, per size.
  python tools/hy_threshold.py [value_bytes]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import gpusorting_amd as g
vb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for log2n in (24, 25, 26, 27):
    n = 1 << log2n
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    dv = torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda") if vb else None
    for name, kw in (("default-lsd", dict(plan=1)), ("two-level", dict(plan=2, position_chains_min_log2=20))):
        s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb, **kw)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for it in range(12):
            g.init_random(dk, 10 + it, 0, dv)
            ev0.record(); s.sort(dk, dv); ev1.record(); torch.cuda.synchronize()
            if it >= 2: ts.append(ev0.elapsed_time(ev1))
        ts.sort()
        print(f"vb={vb} 2^{log2n} {name}: median {ts[len(ts)//2]*1000:.1f} us -> {n/ts[len(ts)//2]/1e6:.1f} GKeys/s {s.last_plan()}", flush=True)
        s.close()
