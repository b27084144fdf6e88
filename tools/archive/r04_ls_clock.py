"""Round 4: shader clocks per phase of the local-sort plan's first kernel (workgroup 7), GPUSORT_LS_EXP=0x10000."""
import os, sys, ctypes as C
os.environ.setdefault("GPUSORT_LS_EXP", "0x10000")
import torch
sys.path.insert(0, ".")
from gpusorting_amd import onesweep as osw, _lib
n = 1 << 28
k0 = torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device="cuda").to(torch.int32)
s = osw.OneSweep(n); s.set_plan(True); s.set_profiling(True)
for r in range(3):
    k = k0.clone(); torch.cuda.synchronize(); s.sort(k); torch.cuda.synchronize()
print("first kernel + transpose ms:", s.get_profile()["pass0"])
SLAB_LS = int(sys.argv[1])
pr = s.get_profile()
print("profile:", {k: round(v, 4) for k, v in pr.items()})
out = (C.c_uint32 * 64)()
_lib.check(_lib.load().gs_debug_read_slab(s._h, SLAB_LS + 920, 48, out, None), "read")
pn = ["claim", "geometry", "run list (fill)", "issue loads", "wait keys + rank", "barrier + digit scan", "stage", "look-back", "gbase + barrier", "scatter"]
for p in range(3):
    v = list(out)[16 * p:16 * p + 16]
    tot = sum(v[:10]) or 1
    print(f"pass {p + 1}: {tot} clk per tile over {v[10]} tiles, workgroup life {v[11] / 100.0:.1f} us")
    for nm, c in zip(pn, v[:10]): print(f"    {nm:24s} {c:7d} clk  {100.0 * c / tot:5.1f} %")
out = (C.c_uint32 * 16)()

_lib.check(_lib.load().gs_debug_read_slab(s._h, SLAB_LS + 900, 16, out, None), "read")
names = ["top barrier", "zero + wait keys + barrier", "rank", "barrier", "digit scan + run row", "stage", "issue next loads + barrier", "write out"]
v = list(out)
tot = sum(v[:8])
for nm, c in zip(names, v[:8]): print(f"  {nm:32s} {c:7d} clk  {100.0 * c / tot:5.1f} %")
print(f"  workgroup 7: entry -> loop {v[11] / 100.0:.1f} us, loop {v[10] / 100.0:.1f} us, loop end -> through {((v[13] - v[12]) & 0xffffffff) / 100.0 - v[11] / 100.0 - v[10] / 100.0:.1f} us; last workgroup ({v[15]}) through {((v[14] - v[12]) & 0xffffffff) / 100.0:.1f} us after workgroup 7's entry")
print(f"  total {tot} clk per tile over {v[8]} tiles; shader clock {v[9] / max(v[10], 1) * 100.0:.0f} MHz ({v[9]} clk in {v[10] / 100.0:.1f} us)")
