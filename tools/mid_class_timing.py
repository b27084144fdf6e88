import os, sys, torch
sys.path.insert(0, os.getcwd())
import gpusorting_amd as g
vb = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for n in ((1 << 21), 3000000, (1 << 22), 6000000, (1 << 23), (1 << 24)):
    dk = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(8)]
    dv = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(8)] if vb else [None] * 8
    for name, kw in (("default", {}), ("mid_path=0", dict(mid_path=0))):
        s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb, **kw)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rnd in range(4):
            for i, t in enumerate(dk): g.init_random(t, 10 + rnd * 8 + i, 0, dv[i])
            torch.cuda.synchronize()
            a.record()
            for t, v in zip(dk, dv): s.sort(t, v)
            b.record(); b.synchronize()
            if rnd: ts.append(a.elapsed_time(b) / len(dk))
        s.check()
        ok = g.validate(dk[-1]) == 0
        ts.sort()
        print(f"vb={vb} n={n} {name}: {ts[len(ts)//2]*1000:.1f} us per sort -> {n/ts[len(ts)//2]/1e6:.1f} GKeys/s sorted={ok}", flush=True)
        s.close()
