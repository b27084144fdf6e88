"""Round 4: who allocates the alternate buffers — the caller (torch's caching allocator) or the library (hipMalloc inside the handle)?
2^28 uniform keys, profile slots.  usage: r04_alt_owner.py"""
import sys, torch
sys.path.insert(0, ".")
import gpusorting_amd as g
n = 1 << 28
for vb in (8, 4, 0):
    vdt = torch.int64 if vb == 8 else torch.int32
    for owner in ("torch", "lib", "torch", "lib"):
        k = torch.empty(n, dtype=torch.int32, device="cuda"); v = torch.empty(n, dtype=vdt, device="cuda") if vb else None
        ka = torch.empty(n, dtype=torch.int32, device="cuda") if owner == "torch" else None
        va = torch.empty(n, dtype=vdt, device="cuda") if (vb and owner == "torch") else None
        s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb); s.set_profiling(True)
        best = None
        for r in range(5):
            g.init_random(k, 10 + r, 0, v); torch.cuda.synchronize()
            if owner == "torch": s.sort(k, v, alt_keys=ka, alt_values=va)
            else: s.sort(k, v)
            torch.cuda.synchronize()
            p = s.get_profile()
            if r and (best is None or p["total"] < best["total"]): best = p
        print(f"vb={vb} alt owner={owner:5s}: total {best['total']:.3f} hist {best['global_histogram']:.3f} passes [{best['pass0']:.3f} {best['pass1']:.3f} {best['pass2']:.3f} {best['pass3']:.3f}]", flush=True)
        s.close(); del k, v, ka, va
