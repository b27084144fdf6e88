#!/usr/bin/env python3
"""On-device tuning sweep: every compiled tile shape x {keys, pairs4, pairs8} at 2^LOG keys.
Prints one line per configuration with GKeys/s and the per-kernel HIP-event breakdown.
Usage: python tools/sweep.py [log2_keys=28] [reps=5] [modes=0,4,8]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402

SHAPES = [(512, 16), (256, 16), (512, 8), (1024, 8), (256, 32), (512, 32), (1024, 16)]


def copy_bw(n):
    a = torch.empty(n, dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    a.random_()
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        b.copy_(a)
    e.record()
    e.synchronize()
    return 8.0 * n * 10 / (s.elapsed_time(e) * 1e-3) / 1e9


def main():
    log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    modes = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,4,8").split(",")]
    n = 1 << log2
    print(f"device: {torch.cuda.get_device_name(0)}  n=2^{log2}  copy(read+write) bandwidth: {copy_bw(n):.0f} GB/s", flush=True)
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    alt = torch.empty_like(keys)
    for vb in modes:
        vdt = torch.int32 if vb == 4 else torch.int64
        vals = torch.empty(n, dtype=vdt, device="cuda") if vb else None
        valt = torch.empty_like(vals) if vb else None
        for (t, k) in SHAPES:
            s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
            try:
                s.set_shape(t, k)
            except g.GpuSortError:
                s.close()
                continue
            s.set_profiling(True)
            acc = {}
            ok = True
            try:
                for r in range(reps + 1):
                    g.init_random(keys, 10 + r, 0, vals)
                    torch.cuda.synchronize()
                    s.sort(keys, vals, alt_keys=alt, alt_values=valt)
                    p = s.get_profile()
                    if r:
                        for kk, v in p.items():
                            acc[kk] = acc.get(kk, 0.0) + v / reps
                s.check()
                ok = g.validate(keys, vals if vb == 4 else None) == 0
            except g.GpuSortError as e:
                print(f"vb={vb} {t}x{k}: ERROR {e}", flush=True)
                s.close()
                continue
            bpk = 4 + 4 * (8 + 2 * vb)
            tot = acc["total"]
            passes = " ".join(f"{acc[f'pass{i}']:.3f}" for i in range(4))
            print(f"vb={vb} {t:4d}x{k:<2d} tile={t*k:5d}  {n/tot/1e6:7.2f} GKeys/s  total={tot:.3f} ms  "
                  f"({bpk*n/tot/1e6/8000*100:4.1f}% of 8TB/s)  clear={acc['clear']:.3f} hist={acc['global_histogram']:.3f} "
                  f"scan={acc['scan']:.3f} passes=[{passes}]  sorted={ok}", flush=True)
            s.close()


if __name__ == "__main__":
    main()
