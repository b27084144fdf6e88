#!/usr/bin/env python3
"""Does the relative placement of the four big buffers of a (u32 key, u64 value) sort matter?  The same sort with the
alt buffers directly behind the inputs (power-of-two spacing) and with gaps between them; argv[2] = realloc: a fresh arena per case."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

n = 1 << 28
POOL = None
REALLOC = len(sys.argv) > 2 and sys.argv[2] == 'realloc'  # a new arena for every case (default: one arena for all)
vb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
vdt = torch.int64 if vb == 8 else torch.int32


def run(tag, shifts):
    """shifts = extra bytes in front of (alt, vals, valt); the arena is 2 MiB aligned, keys sit at its start"""
    if isinstance(shifts, int):
        shifts = (shifts, shifts, shifts)
    global POOL
    if POOL is None or REALLOC:
        POOL = None
        torch.cuda.empty_cache()
        POOL = torch.empty((4 * n + (n * vb) * 2 + 4 * n + (256 << 20)) // 4 + 16, dtype=torch.int32, device="cuda")  # one arena
    pool = POOL
    base = (pool.data_ptr() + (2 << 20) - 1) & ~((2 << 20) - 1)
    off = base - pool.data_ptr()

    def carve(nbytes, dtype, extra=0):
        nonlocal off
        off = (off + extra + 255) & ~255
        t = pool.view(torch.uint8)[off:off + nbytes].view(dtype)
        off += nbytes
        return t
    k = carve(4 * n, torch.int32)
    alt = carve(4 * n, torch.int32, shifts[0])
    v = carve(vb * n, vdt, shifts[1]) if vb else None
    valt = carve(vb * n, vdt, shifts[2]) if vb else None
    s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
    s.set_profiling(True)
    best = None
    for r in range(4):
        g.init_random(k, 10 + r, 0, v)
        s.sort(k, v, alt_keys=alt, alt_values=valt)
        torch.cuda.synchronize()
        p = s.get_profile()
        if r and (best is None or p["total"] < best["total"]):
            best = p
    print(f"vb={vb} {tag:30s} total={best['total']:.3f} ms passes=[{best['pass0']:.3f} {best['pass1']:.3f} {best['pass2']:.3f} {best['pass3']:.3f}]"
          f"  offsets alt/vals/valt - keys = {[hex((t.data_ptr() - k.data_ptr())) for t in (alt, v, valt) if t is not None]}")
    s.close()


K = 1 << 10
cases = [("back to back", 0)]
for sh in (8, 16, 32, 64, 68, 128, 136, 192, 256, 512, 1024, 2048, 2112):
    cases.append((f"+{sh} KiB each", sh * K))
for name, sh in (("alt only +68K", (68 * K, 0, 0)), ("vals only +68K", (0, 68 * K, 0)), ("valt only +68K", (0, 0, 68 * K)),
                 ("alt +68K, valt +136K", (68 * K, 0, 136 * K)), ("alt+64K vals+128K valt+192K", (64 * K, 64 * K, 64 * K)),
                 ("alt+36K vals+72K valt+108K", (36 * K, 36 * K, 36 * K)), ("alt+20K vals+40K valt+60K", (20 * K, 20 * K, 20 * K))):
    cases.append((name, sh))
for tag, sh in cases:
    if vb == 0 and not isinstance(sh, int):
        continue
    run(tag, sh)
