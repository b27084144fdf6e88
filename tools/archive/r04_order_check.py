import sys, torch
sys.path.insert(0, ".")
import gpusorting_amd as g
for n in ((1 << 26) + 12345, 1 << 24, (1<<20)+7):
    for order in (0, 1):
        for preset in (0, 2):
            k = torch.empty(n, dtype=torch.int32, device="cuda")
            g.init_random(k, 3, preset); torch.cuda.synchronize()
            ref = torch.sort(k.to(torch.int64) & 0xffffffff, descending=bool(order)).values
            s = g.OneSweep(n, order=order); s.sort(k); s.check(); torch.cuda.synchronize()
            print(f"n={n} order={order} preset={preset+1} ok={bool(torch.equal(k.to(torch.int64) & 0xffffffff, ref))}", flush=True)
            s.close()
