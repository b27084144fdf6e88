#!/usr/bin/env python3
"""Size sweep 2^10 .. 2^28 and entropy sweep, the reference's BenchmarkOneSweep shape
(GPUSortingD3D12/Tests.h:370-418): keys then pairs.  Usage: size_sweep.py [reps=50]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402


def run(n, vb, entropy, reps):
    vdt = torch.int32 if vb == 4 else torch.int64
    nb = max(1, min(reps, (1 << 30) // (n * 4)))          # distinct pre-generated inputs resident in HBM
    keys = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nb)]
    vals = [torch.empty(n, dtype=vdt, device="cuda") for _ in range(nb)] if vb else [None] * nb
    alt = torch.empty(n, dtype=torch.int32, device="cuda")
    valt = torch.empty(n, dtype=vdt, device="cuda") if vb else None
    s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
    total = 0.0
    done = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while done < reps:
        for i in range(nb):
            g.init_random(keys[i], 10 + done + i, entropy, vals[i])
        torch.cuda.synchronize()
        a.record()
        for i in range(nb):
            s.sort(keys[i], vals[i], alt_keys=alt, alt_values=valt)
        b.record()
        b.synchronize()
        total += a.elapsed_time(b)
        done += nb
    s.check()
    ok = g.validate(keys[-1], vals[-1] if vb == 4 else None) == 0
    s.close()
    return total / done, ok


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    print(f"device: {torch.cuda.get_device_name(0)}; {reps} sorts per point, back-to-back on one stream, inputs pre-generated")
    for vb in (0, 4):
        print(f"--- size sweep, {'keys' if not vb else 'pairs (u32 values)'}, entropy preset 1")
        for lg in range(10, 29):
            ms, ok = run(1 << lg, vb, 0, reps if lg < 26 else max(5, reps // 5))
            print(f"2^{lg:<2d} {ms*1e3:10.1f} us/sort  {(1 << lg)/ms/1e6:9.3f} GKeys/s  sorted={ok}", flush=True)
    for vb in (0, 4, 8):
        print(f"--- entropy sweep at 2^28, value bytes {vb}")
        for e in range(5):
            ms, ok = run(1 << 28, vb, e, 10)
            print(f"preset {e+1} {ms:8.3f} ms/sort  {(1 << 28)/ms/1e6:9.3f} GKeys/s  sorted={ok}", flush=True)


if __name__ == "__main__":
    main()
