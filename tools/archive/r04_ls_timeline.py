"""Round 4: when the workgroups of the local-sort plan's first kernel start and end (100 MHz wall clock; GPUSORT_LS_EXP=0x10000)."""
import os, sys, ctypes as C
os.environ.setdefault("GPUSORT_LS_EXP", "0x10000")
import numpy as np, torch
sys.path.insert(0, ".")
from gpusorting_amd import onesweep as osw, _lib
n = 1 << 28
k0 = torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device="cuda").to(torch.int32)
s = osw.OneSweep(n); s.set_plan(True); s.set_profiling(True)
for r in range(3):
    k = k0.clone(); torch.cuda.synchronize(); s.sort(k); torch.cuda.synchronize()
print("profile:", {k: round(v, 4) for k, v in s.get_profile().items()})
grid, SL = 512, 4608
buf = (C.c_uint32 * (grid * SL))()
_lib.check(_lib.load().gs_debug_read_slab(s._h, 0x80000000, grid * SL, buf, None), "read")
a = np.frombuffer(buf, dtype=np.uint32).reshape(grid, SL)
st, en = a[:, 4352 + 8].astype(np.int64), a[:, 4352 + 9].astype(np.int64)
t0 = st.min()
st, en = (st - t0) / 100.0, (en - t0) / 100.0
print(f"starts: min 0, median {np.median(st):.1f}, max {st.max():.1f} us;  ends: min {en.min():.1f}, median {np.median(en):.1f}, p90 {np.percentile(en, 90):.1f}, max {en.max():.1f} us")
print("mean end by XCD (block % 8):", [round(float(en[x::8].mean()), 1) for x in range(8)])
print("mean life by XCD:", [round(float((en - st)[x::8].mean()), 1) for x in range(8)])
order = np.argsort(en)
print("last 12 to finish (block: start -> end):", [(int(b), round(float(st[b]), 1), round(float(en[b]), 1)) for b in order[-12:]])
print("first 6 to finish:", [(int(b), round(float(st[b]), 1), round(float(en[b]), 1)) for b in order[:6]])
h, edges = np.histogram(en, bins=12)
print("end histogram:", list(zip([round(float(e), 0) for e in edges[:-1]], h.tolist())))
