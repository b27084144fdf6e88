#!/usr/bin/env python3
"""Round 3: where a mid-size sort spends its time, kernel by kernel, gaps included.
  run  <log2n> [vb=0] [reps=3]   back-to-back sorts of distinct buffers (what rocprofv3 --kernel-trace --output-format csv wraps)
  parse <kernel_trace.csv>        per kernel of the sort: mean duration and mean idle time in front of it, over the sorts of the
                                  trace's second half (the first half warms up)"""
import csv
import os
import sys
from collections import OrderedDict

SORT_KERNELS = ("global_histogram", "hist_reduce", "scan_kernel", "digit_binning", "mid_msd", "bucket_sort", "small_sort")


def run(lg, vb, reps):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gpusorting_amd as g
    n = 1 << lg
    nb = max(2, min(16, (1 << 30) // (n * 4)))
    ks = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nb)]
    vs = [torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda") for _ in range(nb)] if vb else [None] * nb
    s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(reps):
        for i in range(nb):
            g.init_random(ks[i], 10 + i + 100 * rep, 0, vs[i])
        torch.cuda.synchronize()
        a.record()
        for i in range(nb):
            s.sort(ks[i], vs[i])
        b.record()
        b.synchronize()
        print(f"2^{lg} vb={vb} rep {rep}: {a.elapsed_time(b) / nb * 1e3:.1f} us per sort", flush=True)
    s.check()
    s.close()


def parse(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = [(a, b, k) for a, b, k in rows]
    half = rows[len(rows) // 2:]
    seq = OrderedDict()
    prev_end, nsorts, first_start, span = None, 0, None, 0
    pos = 0
    for a, b, k in half:
        short = next((s for s in SORT_KERNELS if s in k), None)
        if short is None:
            prev_end = None
            continue
        if short in ("global_histogram", "mid_msd", "small_sort"):
            pos = 0
            if prev_end is not None and first_start is not None:
                nsorts += 1
                span += a - first_start
            first_start = a
        key = f"{pos}:{short}"
        e = seq.setdefault(key, [0, 0, 0, 0])
        e[0] += 1
        e[1] += b - a
        if prev_end is not None:
            e[2] += a - prev_end
            e[3] += 1
        prev_end = b
        pos += 1
    print(f"== {path}: {nsorts} back-to-back sorts, {span / max(nsorts, 1) / 1e3:.1f} us from one sort's first kernel to the next one's")
    for key, (c, dur, gap, gc) in seq.items():
        print(f"  {key:24s} calls {c:4d}  run {dur / c / 1e3:7.2f} us   idle before {gap / max(gc, 1) / 1e3:6.2f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0, int(sys.argv[4]) if len(sys.argv) > 4 else 3)
    else:
        for p in sys.argv[2:]:
            parse(p)
