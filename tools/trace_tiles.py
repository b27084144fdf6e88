#!/usr/bin/env python3
"""Per-tile phase timing of the DigitBinningPass (needs the GS_EXP=2 build:
GPUSORT_LIB=gpusorting_amd/lib/libgpusort_trace.so).  Usage: trace_tiles.py [log2=28] [TxK=512x32] [entropy_preset_index=0] [value_bytes=0]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402
from gpusorting_amd import _lib  # noqa: E402


def main():
    a = sys.argv[1:]
    log2 = int(a[0]) if a else 28
    t, k = (int(x) for x in (a[1] if len(a) > 1 else "512x32").split("x"))
    preset = int(a[2]) if len(a) > 2 else 0
    vb = int(a[3]) if len(a) > 3 else 0
    n = 1 << log2
    lib = _lib.load()
    lib.gs_debug_set_trace.argtypes = [C.c_void_p, C.c_void_p]
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    alt = torch.empty_like(keys)
    s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
    vals = None if not vb else torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
    s.set_shape(t, k)
    grid = (n + t * k - 1) // (t * k) + 34  # tiles + MAXCH + 1 (gpusort_capi.hip prologue)
    tr = torch.zeros(4 * grid * 8, dtype=torch.int32, device="cuda")
    lib.gs_debug_set_trace(s._h, tr.data_ptr())
    for rep in range(2):
        g.init_random(keys, 10 + rep, preset, vals)
        torch.cuda.synchronize()
        s.sort(keys, vals)
        s.check()
    d = tr.cpu().numpy().view(np.uint32).reshape(4, grid, 8).astype(np.int64)
    names = ["claim", "load+rank", "reduce+RED+fold", "stage", "lookback", "scatter", "tile total"]
    print(f"entropy preset {preset + 1}; shape {t}x{k} blocks/pass={grid} (10 ns ticks -> us; thread 0 of each workgroup)")
    for p in range(4):
        x = d[p]
        x = x[(x[:, 7] >> 31) == 1]
        ts = x[:, :7]
        span = (ts[:, 6].max() - ts[:, 0].min()) / 100.0
        ph = [ts[:, 1] - ts[:, 0], ts[:, 2] - ts[:, 1], ts[:, 3] - ts[:, 2], ts[:, 4] - ts[:, 3], ts[:, 5] - ts[:, 4],
              ts[:, 6] - ts[:, 5], ts[:, 6] - ts[:, 0]]
        trips = x[:, 7] & 0xffff
        print(f"pass {p}: tiles {len(x)} span {span:.1f} us; look-back trips mean {trips.mean():.2f} p99 {np.percentile(trips,99):.0f} max {trips.max()}")
        st = (ts[:, 0] - ts[:, 0].min()) / 100.0
        en = (ts[:, 6] - ts[:, 0].min()) / 100.0
        print(f"    tile start after the first: p50 {np.median(st):.2f} p90 {np.percentile(st,90):.2f} max {st.max():.2f} us;  tile end: p10 {np.percentile(en,10):.2f} p50 {np.median(en):.2f} max {en.max():.2f} us")
        for nm, v in zip(names, ph):
            v = v / 100.0
            print(f"    {nm:16s} mean {v.mean():6.2f}  p50 {np.median(v):6.2f}  p90 {np.percentile(v,90):6.2f}  p99 {np.percentile(v,99):6.2f}  max {v.max():7.2f} us")
    s.close()


if __name__ == "__main__":
    main()
