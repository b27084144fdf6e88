import os, sys, torch
sys.path.insert(0, os.getcwd())
import gpusorting_amd as g
for vb in (4, 8):
    for shape in ("auto", "1024x16" if vb == 4 else "512x32", "512x16"):
        row = []
        for lg in range(18, 26):
            n = 1 << lg
            vdt = torch.int32 if vb == 4 else torch.int64
            nb = 8
            keys = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nb)]
            vals = [torch.empty(n, dtype=vdt, device="cuda") for _ in range(nb)]
            s = g.OneSweep(n, mode=g.MODE_PAIRS, value_bytes=vb)
            if shape != "auto":
                t, k = (int(x) for x in shape.split("x")); s.set_shape(t, k)
            best = 1e9
            for rep in range(3):
                for i in range(nb): g.init_random(keys[i], 10 + i + rep, 0, vals[i])
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(nb): s.sort(keys[i], vals[i])
                b.record(); b.synchronize()
                best = min(best, a.elapsed_time(b) / nb * 1e3)
            row.append(f"{best:7.1f}")
            s.close()
        print(f"vb={vb} {shape:8s}", " ".join(row), flush=True)
