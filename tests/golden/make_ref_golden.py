#!/usr/bin/env python3
"""Generates tests/golden/ref_init_random.npz from the REFERENCE's OWN input generator.

oracle/_ref/libref_generator.so (oracle/Makefile `_ref`) is the reference's
GPUSortingCUDA/UtilityKernels.cuh:53-117 compiled for the CPU from where it lies under /root/reference and run
one emulated CUDA thread at a time with the reference's launch shape <<<256,256>>>.  This script can therefore
only run where /root/reference exists; its output is committed so that the oracle's restatement and the HIP
generator are checked against reference-produced keys on boxes that have no reference tree.

Per case (n, seed, andCount): crc32 of all keys, and the first / last 64 keys verbatim.
Run:  make -C oracle _ref && python tests/golden/make_ref_golden.py
"""
import ctypes as C
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = [(1, 1, 0), (2, 7, 0), (63, 2, 0), (7680, 7680, 0), (11111, 11111, 1), (15360, 15360, 0), (65535, 3, 2),
         (65536, 10, 0), (65537, 10, 1), (200003, 77, 4), (1 << 20, 26, 0), (1 << 20, 27, 3), ((1 << 22) + 5, 28, 2)]


def main():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_generator.so"))
    ref.ref_init_random.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    ref.ref_generator_source.restype = C.c_char_p
    out = {"cases": np.array(CASES, dtype=np.int64), "source": np.array(ref.ref_generator_source().decode())}
    for i, (n, seed, andc) in enumerate(CASES):
        k = np.empty(n, np.uint32)
        p = np.empty(n, np.uint32)
        ref.ref_init_random(k.ctypes.data, p.ctypes.data, andc, seed, n)
        assert np.array_equal(k, p)  # the pairs overload writes payload = key (UtilityKernels.cuh:114-115)
        k2 = np.empty(n, np.uint32)
        ref.ref_init_random(k2.ctypes.data, None, andc, seed, n)
        assert np.array_equal(k, k2)
        out[f"crc_{i}"] = np.uint32(zlib.crc32(k.tobytes()) & 0xFFFFFFFF)
        out[f"head_{i}"] = k[:64].copy()
        out[f"tail_{i}"] = k[-64:].copy()
    path = os.path.join(HERE, "ref_init_random.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(CASES), "cases")


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------------
# Part 2: the reference's OneSweep KERNELS (GPUSortingCUDA/Sort/OneSweep.cu) executed by the SIMT emulator
# (oracle/_ref/libref_onesweep.so, oracle/ref_onesweep.cpp).  Per case: the error count of the reference's
# Validate kernel on the unsorted input (and the check that it is 0 on the output), crc32 of the global histogram, of the
# key buffer after each of the four passes (the fourth = the sorted result) and, for pairs, of the payload
# buffer after each pass; plus the first/last 32 sorted keys verbatim.  Inputs come from the reference's own
# generator (part 1); payload = original index, which exposes stability.
SORT_CASES = [  # (n, seed, andCount, pairs)
    (1, 5, 0, 0), (2, 5, 0, 1), (31, 9, 0, 0), (4096, 4096, 0, 1), (7679, 7679, 0, 0), (7680, 7680, 0, 0),
    (7681, 7681, 0, 1), (9000, 9000, 2, 1), (11111, 11111, 0, 0), (15359, 15359, 4, 1), (15360, 15360, 0, 0),
    (15361, 15361, 0, 0), (40000, 77, 3, 1), (65536, 10, 0, 0), (100003, 12, 1, 1), ((1 << 17) + 3, 26, 0, 0),
]


def sort_goldens():
    gen = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_generator.so"))
    gen.ref_init_random.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_onesweep.so"))
    ref.ref_onesweep_sort_keys.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    ref.ref_onesweep_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    ref.ref_onesweep_source.restype = C.c_char_p
    ref.ref_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    ref.ref_validate.restype = C.c_uint32
    crc = lambda a: np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF)
    out = {"cases": np.array(SORT_CASES, dtype=np.int64), "source": np.array(ref.ref_onesweep_source().decode())}
    for i, (n, seed, andc, pairs) in enumerate(SORT_CASES):
        k = np.empty(n, np.uint32)
        gen.ref_init_random(k.ctypes.data, None, andc, seed, n)
        # the reference's Validate kernel (UtilityKernels.cuh:402-479) on the UNSORTED input: its error count
        out[f"verr_{i}"] = np.uint32(ref.ref_validate(k.ctypes.data, None, n))
        gh = np.zeros(1024, np.uint32)
        ap = np.zeros(4 * n, np.uint32)
        if pairs:
            v = np.arange(n, dtype=np.uint32)
            vp = np.zeros(4 * n, np.uint32)
            ref.ref_onesweep_sort_pairs(k.ctypes.data, v.ctypes.data, n, gh.ctypes.data, ap.ctypes.data, vp.ctypes.data)
            assert np.array_equal(v, vp[3 * n:])
            out[f"vcrc_{i}"] = np.array([crc(vp[p * n:(p + 1) * n]) for p in range(4)], dtype=np.uint32)
        else:
            ref.ref_onesweep_sort_keys(k.ctypes.data, n, gh.ctypes.data, ap.ctypes.data)
        assert np.array_equal(k, ap[3 * n:]) and bool(np.all(k[1:] >= k[:-1]))
        assert ref.ref_validate(k.ctypes.data, None, n) == 0  # the reference's own pass criterion on its own output
        out[f"hist_{i}"] = crc(gh)
        out[f"kcrc_{i}"] = np.array([crc(ap[p * n:(p + 1) * n]) for p in range(4)], dtype=np.uint32)
        out[f"head_{i}"] = k[:32].copy()
        out[f"tail_{i}"] = k[-32:].copy()
        print(f"  reference kernels: n={n} seed={seed} and={andc} pairs={pairs} ok", flush=True)
    path = os.path.join(HERE, "ref_onesweep.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(SORT_CASES), "cases")


if __name__ == "__main__":
    sort_goldens()
