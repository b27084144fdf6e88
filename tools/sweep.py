#!/usr/bin/env python3
"""On-device tuning sweep: every compiled tile shape x rank mode x {keys, pairs4, pairs8} at 2^LOG keys.
Prints one line per configuration with GKeys/s and the per-kernel HIP-event breakdown.
Usage: python tools/sweep.py [log2_keys=28] [reps=5] [modes=0,4,8] [ranks=0,1] [entropy=0]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402
from gpusorting_amd import _lib  # noqa: E402

SHAPES = [(512, 32), (512, 20), (1024, 16), (512, 16), (256, 32), (256, 16)]


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


def main():
    a = sys.argv[1:]
    log2 = int(a[0]) if len(a) > 0 else 28
    reps = int(a[1]) if len(a) > 1 else 5
    modes = [int(x) for x in (a[2] if len(a) > 2 else "0,4,8").split(",")]
    ranks = [int(x) for x in (a[3] if len(a) > 3 else "0,1").split(",")]
    entropy = int(a[4]) if len(a) > 4 else 0

    n = 1 << log2
    lib = _lib.load()
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    alt = torch.empty_like(keys)
    keys.random_()
    ms = timed(lambda: alt.copy_(keys))
    print(f"device: {torch.cuda.get_device_name(0)}  n=2^{log2}  lib={_lib.LIB_PATH}")
    print(f"torch copy: {ms:.3f} ms = {8.0*n/ms/1e6:.0f} GB/s (read+write)", flush=True)
    sp = int(torch.cuda.current_stream().cuda_stream)
    for (t, k) in ((512, 16), (1024, 16), (256, 32)):
        ms = timed(lambda: _lib.load_tuning().gs_debug_copy_floor(keys.data_ptr(), alt.data_ptr(), n, t, k, sp))
        print(f"copy_floor {t}x{k}: {ms:.3f} ms = {8.0*n/ms/1e6:.0f} GB/s", flush=True)
    fails = C.c_uint64(1)
    st = lib.gs_selftest_lds_atomic_order(2000, 1, C.byref(fails), sp)
    print(f"lds atomic lane-order probe: status={st} failures={fails.value} "
          f"(of {1024*8*2000*8*64} ranked lanes)", flush=True)
    for vb in modes:
        vdt = torch.int32 if vb == 4 else torch.int64
        vals = torch.empty(n, dtype=vdt, device="cuda") if vb else None
        valt = torch.empty_like(vals) if vb else None
        for (t, k) in SHAPES:
            for rank in ranks:
                s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
                try:
                    s.set_shape(t, k)
                    s.set_rank_mode(rank)
                except g.GpuSortError:
                    s.close()
                    continue
                s.set_profiling(True)
                acc = {}
                try:
                    for r in range(reps + 1):
                        g.init_random(keys, 10 + r, entropy, vals)
                        torch.cuda.synchronize()
                        s.sort(keys, vals, alt_keys=alt, alt_values=valt)
                        p = s.get_profile()
                        if r:
                            for kk, v in p.items():
                                acc[kk] = acc.get(kk, 0.0) + v / reps
                    s.check()
                    ok = g.validate(keys, vals if vb == 4 else None) == 0
                except g.GpuSortError as e:
                    print(f"vb={vb} {t}x{k} rank={rank}: ERROR {e}", flush=True)
                    s.close()
                    continue
                bpk = 4 + 4 * (8 + 2 * vb)
                tot = acc["total"]
                passes = " ".join(f"{acc[f'pass{i}']:.3f}" for i in range(4))
                print(f"vb={vb} {t:4d}x{k:<2d} rank={rank} tile={t*k:5d}  {n/tot/1e6:7.2f} GKeys/s  total={tot:.3f} ms  "
                      f"({bpk*n/tot/1e6/8000*100:4.1f}% of 8TB/s)  clear={acc['clear']:.3f} hist={acc['global_histogram']:.3f} "
                      f"scan={acc['scan']:.3f} passes=[{passes}]  sorted={ok}", flush=True)
                s.close()


if __name__ == "__main__":
    main()
