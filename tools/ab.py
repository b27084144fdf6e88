#!/usr/bin/env python3
"""A/B of library builds on one box: best-of-N whole-sort time per (library, value bytes), libraries interleaved.
Usage: ab.py lib1.so lib2.so ... [--log2 28] [--rounds 4] [--vb 0,4,8] [--preset 0]"""
import os
import subprocess
import sys

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
opt = {"--log2": "28", "--rounds": "4", "--vb": "0,4,8", "--preset": "0"}
for i, a in enumerate(sys.argv):
    if a in opt:
        opt[a] = sys.argv[i + 1]
child = r'''
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1]))))
import gpusorting_amd as g
n = 1 << int(sys.argv[2]); preset = int(sys.argv[4])
for vb in [int(x) for x in sys.argv[3].split(",")]:
    k = torch.empty(n, dtype=torch.int32, device="cuda")
    v = None if not vb else torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
    s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
    s.set_profiling(True)
    best = None
    for r in range(6):
        g.init_random(k, 10 + r, preset, v)
        s.sort(k, v); torch.cuda.synchronize()
        p = s.get_profile()
        if r and (best is None or p["total"] < best["total"]): best = p
    print(vb, best["total"], best["pass0"], best["pass1"], best["pass2"], best["pass3"], best["global_histogram"])
    s.close()
'''
res = {}
here = os.path.abspath(__file__)
for rnd in range(int(opt["--rounds"])):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", child, here, opt["--log2"], opt["--vb"], opt["--preset"]],
                             env=dict(os.environ, GPUSORT_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        for line in out.stdout.splitlines():
            f = line.split()
            if len(f) == 7:
                key = (lib, int(f[0]))
                vals = [float(x) for x in f[1:]]
                if key not in res or vals[0] < res[key][0]:
                    res[key] = vals
        if out.returncode:
            print(lib, "FAILED", out.stderr[-300:])
for (lib, vb), v in sorted(res.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"vb={vb} {os.path.basename(lib):32s} total={v[0]:.3f} ms passes=[{v[1]:.3f} {v[2]:.3f} {v[3]:.3f} {v[4]:.3f}] hist={v[5]:.3f}")
