import os, sys, torch
sys.path.insert(0, os.getcwd())
import gpusorting_amd as g
for shape in ("512x32", "512x16", "256x32", "256x16"):
    t, k = (int(x) for x in shape.split("x"))
    row = []
    for lg in range(14, 24):
        n = 1 << lg
        keys = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(20)]
        alt = torch.empty(n, dtype=torch.int32, device="cuda")
        s = g.OneSweep(n); s.set_shape(t, k)
        best = 1e9
        for rep in range(3):
            for i, kk in enumerate(keys): g.init_random(kk, 10 + i + rep, 0)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for kk in keys: s.sort(kk, alt_keys=alt)
            b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b) / len(keys) * 1e3)
        row.append(f"{best:6.1f}")
        s.close()
    print(shape, " ".join(row), flush=True)
