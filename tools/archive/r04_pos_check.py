import sys, torch
sys.path.insert(0, ".")
import gpusorting_amd as g
for lg, extra in ((26, 777), (28, 0), (25, 1)):
    n = (1 << lg) + extra
    for preset in (1, 2, 3, 4):
        k = torch.empty(n, dtype=torch.int32, device="cuda")
        g.init_random(k, 5 + preset, preset); torch.cuda.synchronize()
        want = torch.sort(k.view(torch.uint32).to(torch.int64)).values if False else None
        ref = torch.sort((k.to(torch.int64) & 0xffffffff)).values
        s = g.OneSweep(n); s.sort(k); s.check(); torch.cuda.synchronize()
        ok = bool(torch.equal(k.to(torch.int64) & 0xffffffff, ref))
        print(f"n={n} preset={preset+1} ok={ok}", flush=True)
        s.close(); del k, ref
