#!/usr/bin/env python3
"""Randomized cross-check (CPU only, needs oracle/_ref, i.e. /root/reference at build time): the reference's OneSweep
kernels under the SIMT emulator vs the oracle — global histogram, the buffer after every pass, the result, stable
payloads — on random sizes / entropy presets / distributions for ~150 s.  Usage: python tools/ref_vs_oracle_sweep.py"""
import ctypes as C, numpy as np, sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import oracle_lib
o = oracle_lib.load()
ref = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'libref_onesweep.so'))
ref.ref_onesweep_sort_keys.argtypes=[C.c_void_p,C.c_uint32,C.c_void_p,C.c_void_p]
ref.ref_onesweep_sort_pairs.argtypes=[C.c_void_p,C.c_void_p,C.c_uint32,C.c_void_p,C.c_void_p,C.c_void_p]
rng = np.random.default_rng(2026)
bad = 0; t0 = time.time(); cases = 0
while time.time() - t0 < 150:
    n = int(rng.integers(1, 50000)); andc = int(rng.integers(0, 5)); seed = int(rng.integers(1, 1 << 31))
    kind = int(rng.integers(0, 4))
    keys = o.init_random(n, seed, andc)
    if kind == 1: keys = (keys % np.uint32(int(rng.integers(1, 300)))).astype(np.uint32)
    if kind == 2: keys = np.sort(keys)[::-1].copy()
    if kind == 3: keys = (keys & np.uint32(0x00FF00FF)).astype(np.uint32)
    k = keys.copy(); gh = np.zeros(1024, np.uint32); ap = np.zeros(4*n, np.uint32)
    ref.ref_onesweep_sort_keys(k.ctypes.data, n, gh.ctypes.data, ap.ctypes.data)
    ok = np.array_equal(k, o.std_sort(keys)) and np.array_equal(gh.reshape(4,256), o.global_histogram(keys))
    cur = keys
    for p in range(4):
        cur = o.digit_pass(cur, 8*p); ok = ok and np.array_equal(ap[p*n:(p+1)*n], cur)
    k2 = keys.copy(); v2 = np.arange(n, dtype=np.uint32)
    ref.ref_onesweep_sort_pairs(k2.ctypes.data, v2.ctypes.data, n, None, None, None)
    rk, rv = o.std_sort(keys, 0, 0, np.arange(n, dtype=np.uint32))
    ok = ok and np.array_equal(k2, rk) and np.array_equal(v2, rv)
    cases += 1; bad += (not ok)
    if not ok: print("MISMATCH", n, andc, seed, kind)
print(f"{cases} random cases (reference kernels under emulation vs oracle: histogram, every pass, result, stable payloads): {bad} mismatches")
