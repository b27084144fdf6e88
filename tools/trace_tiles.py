#!/usr/bin/env python3
"""Per-tile phase timing of the DigitBinningPass (needs the GS_EXP=2 build:
GPUSORT_LIB=gpusorting_amd/lib/libgpusort_trace.so).  Usage: trace_tiles.py [log2=28] [TxK=512x32] [rank=1]"""
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
    rank = int(a[2]) if len(a) > 2 else 1
    n = 1 << log2
    lib = _lib.load()
    lib.gs_debug_set_trace.argtypes = [C.c_void_p, C.c_void_p]
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    alt = torch.empty_like(keys)
    s = g.OneSweep(n)
    s.set_shape(t, k)
    s.set_rank_mode(rank)
    tiles = (n + t * k - 1) // (t * k)
    tr = torch.zeros(4 * tiles * 8, dtype=torch.int32, device="cuda")
    lib.gs_debug_set_trace(s._h, tr.data_ptr())
    for rep in range(2):
        g.init_random(keys, 10 + rep, 0)
        torch.cuda.synchronize()
        s.sort(keys, alt_keys=alt)
        s.check()
    d = tr.cpu().numpy().view(np.uint32).reshape(4, tiles, 8).astype(np.int64)
    print(f"shape {t}x{k} rank={rank} tiles={tiles}  (10 ns ticks -> us)")
    for p in range(4):
        x = d[p]
        ts = x[:, :6]
        t0 = ts[:, 0].min()
        span = (ts[:, 5].max() - t0) / 100.0
        ph = {
            "load+rank": (ts[:, 1] - ts[:, 0]), "reduce+RED": (ts[:, 2] - ts[:, 1]), "fold+stage": (ts[:, 3] - ts[:, 2]),
            "lookback": (ts[:, 4] - ts[:, 3]), "scatter": (ts[:, 5] - ts[:, 4]), "tile total": (ts[:, 5] - ts[:, 0]),
        }
        print(f"pass {p}: kernel span {span:.1f} us; trips mean {x[:,6].mean():.2f} max {x[:,6].max()}; "
              f"rows walked mean {(x[:,7]&0xffff).mean():.2f} max {(x[:,7]&0xffff).max()}")
        for name, v in ph.items():
            v = v / 100.0
            print(f"    {name:11s} mean {v.mean():7.2f}  p50 {np.median(v):7.2f}  p90 {np.percentile(v,90):7.2f}  "
                  f"p99 {np.percentile(v,99):7.2f}  max {v.max():7.2f} us")
        # start-order vs ticket: how far ahead of tile t-1 does tile t publish its RED?
        red = ts[:, 2] / 100.0
        lag = red[:-1] - red[1:]  # >0: predecessor published later than me
        print(f"    RED(t-1) - RED(t): mean {lag.mean():.2f} p90 {np.percentile(lag,90):.2f} p99 {np.percentile(lag,99):.2f} max {lag.max():.2f} us")
        start = ts[:, 0] / 100.0
        print(f"    start(t) - start(t-1): mean {(start[1:]-start[:-1]).mean():.3f} min {(start[1:]-start[:-1]).min():.2f} max {(start[1:]-start[:-1]).max():.2f} us")
        xcc = (x[:, 7] >> 16) & 0xf
        print(f"    xcc of tiles 0..15: {xcc[:16].tolist()}")
    s.close()


if __name__ == "__main__":
    main()
