#!/usr/bin/env python3
"""Round 3: where the DigitBinningPass loses its 0.06-0.08 ms against the tile-shaped copy — the look-back WAIT or the
SCATTER SHAPE?  Needs a -DGS_EXP=1024 build (GPUSORT_LIB), whose kernels take the variants as runtime mode bits:
   256  replay     descriptors of an identical earlier sort are still there, nobody publishes REDUCTION: every look-back
                   ends in its first read (one round trip), positions exact
   512  early      the predecessor's row is requested before the key loads (with 256: a look-back that never waits)
  1024  sequential output positions tile_base + i (the real look-back still runs)
Also: GlobalHistogram with 1 / 2 / 4 digit tables, and the shader clock under the histogram's load.
Usage: GPUSORT_LIB=gpusorting_amd/lib/libgpusort_exp1024.so python tools/r03_ablate.py [log2n=28] [reps=5] [preset=0]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402
from gpusorting_amd import _lib  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
preset = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 1 << log2n
lib = _lib.load()
print(f"# lib={_lib.LIB_PATH} n=2^{log2n} preset={preset + 1} reps={reps} device={torch.cuda.get_device_name(0)}")
k = torch.empty(n, dtype=torch.int32, device="cuda")
s = g.OneSweep(n)
s.set_profiling(True)


structured = os.environ.get("R03_STRUCTURED", "0") == "1"
master = None
if structured:
    # every 16 384-key tile holds every value of byte 0 exactly 64 times and every wave-load 64 distinct values: the runs of
    # pass 0 are exactly 64 keys long and 256-byte aligned — 256 output streams per chain without a single partial line
    idx = torch.arange(n, dtype=torch.int64, device="cuda")
    hi = torch.randint(0, 1 << 24, (n,), dtype=torch.int64, device="cuda")
    master = ((hi << 8) | (idx & 255)).to(torch.int32)
    del idx, hi
    print("# STRUCTURED input: byte 0 = index & 255 (aligned 64-key runs in pass 0), upper bytes random")


def fill(seed):
    if master is not None:
        k.copy_(master)
    else:
        g.init_random(k, seed, preset)


def one(expmode, seed):
    if os.environ.get("R03_NOPRIME", "0") != "1":
        os.environ["GPUSORT_EXPMODE"] = "0"
        fill(seed)
        s.sort(k)  # primes the descriptors with this input's exact prefixes
        torch.cuda.synchronize()
    os.environ["GPUSORT_EXPMODE"] = str(expmode)
    fill(seed)
    s.sort(k)
    torch.cuda.synchronize()
    p = s.get_profile()
    ok = g.validate(k) == 0 if not (expmode & 1024) else None
    os.environ["GPUSORT_EXPMODE"] = "0"
    return p, ok


names = {0: "real look-back, real scatter (generic scatter code)", 1024: "real look-back, SEQUENTIAL output",
         256: "replay (1 read), real scatter", 256 | 512: "replay + early read (no wait), real scatter",
         512: "real look-back + early first read, real scatter",
         256 | 512 | 1024: "replay + early read, SEQUENTIAL output (= tile machinery floor)",
         256 | 1024: "replay (1 read), SEQUENTIAL output"}
modes = [int(x) for x in os.environ["R03_MODES"].split(",")] if os.environ.get("R03_MODES") else [0, 1024, 256, 256 | 512, 512, 256 | 1024, 256 | 512 | 1024, 0]
for mode in modes:
    best = None
    oks = []
    for r in range(reps):
        p, ok = one(mode, 10 + r)
        oks.append(ok)
        if best is None or p["total"] < best["total"]:
            best = p
    print(f"mode {mode:4d}  total={best['total']:.3f} hist={best['global_histogram']:.3f} passes=[{best['pass0']:.3f} {best['pass1']:.3f} "
          f"{best['pass2']:.3f} {best['pass3']:.3f}] sorted={oks}  # {names.get(mode, '')}")
    sys.stdout.flush()

if os.environ.get("R03_MODES_ONLY", "0") == "1":
    sys.exit(0)
# shader clock under the histogram kernel's load (s_memtime against the 100 MHz s_memrealtime), from block 0
try:
    fn = lib.gs_debug_read_status_words
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]
    g.init_random(k, 10, preset)
    s.sort(k)
    torch.cuda.synchronize()
    w = (C.c_uint32 * 32)()
    fn(s._h, w, None)
    c = [w[16 + 2 * i] | (w[17 + 2 * i] << 32) for i in range(4)]
    dc, dw = c[2] - c[0], c[3] - c[1]
    print(f"histogram kernel block 0: clock64 delta {dc}, wall_clock64 (100 MHz) delta {dw} -> {dw / 100.0:.1f} us, "
          f"clock64 rate {dc / max(dw, 1) * 100.0:.0f} MHz")
except Exception as e:  # noqa: BLE001
    print("clock read failed:", e)

# GlobalHistogram with fewer digit tables: stand-alone pass (np = 1), fine MSD histogram (np = 2), full (np = 4)
k2 = torch.empty(n, dtype=torch.int32, device="cuda")
for label, fn2 in (("np=1 (digit_pass 0)", lambda: s.digit_pass(k, k2, 0)), ("np=1 (digit_pass 2)", lambda: s.digit_pass(k, k2, 2)),
                   ("np=2 (msd_fine_histogram)", lambda: s.msd_fine_histogram(k)), ("np=4 (sort)", lambda: s.sort(k))):
    best = None
    for r in range(reps):
        g.init_random(k, 10 + r, preset)
        torch.cuda.synchronize()
        fn2()
        torch.cuda.synchronize()
        try:
            p = s.get_profile()
        except Exception:  # msd_fine_histogram records no pass events
            p = None
        if p and (best is None or p["global_histogram"] < best["global_histogram"]):
            best = p
    if best:
        print(f"hist {label:28s} global_histogram={best['global_histogram']:.3f} ms  first pass after it={best['pass0']:.3f} ms")
    else:
        print(f"hist {label:28s} (no profile)")
