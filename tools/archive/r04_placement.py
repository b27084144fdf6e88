"""Round 4: does the relative placement of the four buffers of a pairs sort matter?  2^28 (u32, u64) / (u32, u32) pairs, uniform keys;
every buffer is a view into one oversized allocation, shifted by a chosen number of bytes.  usage: r04_placement.py VB"""
import sys, torch
sys.path.insert(0, ".")
import gpusorting_amd as g
vb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1 << 28
pad = 64 << 20
vdt = torch.int64 if vb == 8 else torch.int32
def buf(nbytes): return torch.empty(nbytes + pad, dtype=torch.uint8, device="cuda")
raw = [buf(n * 4), buf(n * vb) if vb else None, buf(n * 4), buf(n * vb) if vb else None]
def view(r, off, dt, cnt):
    return r[off:off + cnt * dt.itemsize].view(dt)
s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb); s.set_profiling(True)
KB, MB = 1 << 10, 1 << 20
cases = [(0, 0, 0, 0), (0, 256, 512, 768), (0, 4 * KB, 8 * KB, 12 * KB), (0, 64 * KB, 128 * KB, 192 * KB), (0, MB, 2 * MB, 3 * MB),
         (0, 2 * MB + 4 * KB, 4 * MB + 8 * KB, 6 * MB + 12 * KB), (0, 0, 16 * MB + 256, 16 * MB + 256), (0, 8 * MB, 16 * MB, 24 * MB),
         (0, 0, 32 * KB, 32 * KB), (0, 16 * KB, 32 * KB, 48 * KB), (0, 1 * KB, 2 * KB, 3 * KB)]
for offs in cases:
    k = view(raw[0], offs[0], torch.int32, n); v = view(raw[1], offs[1], vdt, n) if vb else None
    ka = view(raw[2], offs[2], torch.int32, n); va = view(raw[3], offs[3], vdt, n) if vb else None
    best = None
    for r in range(4):
        g.init_random(k, 10 + r, 0, v); torch.cuda.synchronize()
        s.sort(k, v, alt_keys=ka, alt_values=va); torch.cuda.synchronize()
        p = s.get_profile()
        if r and (best is None or p["total"] < best["total"]): best = p
    print(f"vb={vb} offsets(k,v,ka,va)={offs}: total {best['total']:.3f} passes [{best['pass0']:.3f} {best['pass1']:.3f} {best['pass2']:.3f} {best['pass3']:.3f}]"
          f"  ptr%2MiB: {[hex(t.data_ptr() % (2 * MB)) for t in (k, v, ka, va) if t is not None]}", flush=True)
