#!/usr/bin/env python3
"""Generates tests/golden/onesweep_golden.npz — an INDEPENDENT numpy restatement of
the reference's seeded input generator and of the sort's defined result.

The reference (b0nes164/GPUSorting) ships no golden vectors for this path and
its sort kernels cannot be built here (CUDA/PTX, D3D12, Unity), so these vectors
do not come from the reference itself (the generator's reference-produced
vectors are in ref_init_random.npz, see make_ref_golden.py); they pin the C++
oracle (oracle/gs_oracle.cpp) and the HIP kernels against a second
implementation written from the same spec:
  generator  GPUSortingCUDA/UtilityKernels.cuh:29-33,53-117 (<<<256,256>>>)
  result     stable LSD radix sort == np.sort(kind="stable") on the sortable
             bits; descending == exact reverse (SortCommon.hlsl:594-597)
Run:  python tests/golden/make_golden.py
"""
import os
import zlib

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def init_random_np(n, seed, and_count):
    """65536 virtual threads, one discarded step, and_count+1 AND-ed draws per key."""
    nt = min(n, 65536)
    idx = np.arange(nt, dtype=np.uint64)
    seed = np.uint64(seed & 0xFFFFFFFF)
    z1 = ((idx << np.uint64(2)) & M32) * seed & M32
    z2 = (((idx << np.uint64(2)) + np.uint64(1)) & M32) * seed & M32
    z3 = (((idx << np.uint64(2)) + np.uint64(2)) & M32) * seed & M32
    z4 = (((idx << np.uint64(2)) + np.uint64(3)) & M32) * seed & M32

    def step(z1, z2, z3, z4):
        u = np.uint64
        z1 = (((z1 & u(4294967294)) << u(12)) & M32) ^ ((((z1 << u(13)) & M32) ^ z1) >> u(19))
        z2 = (((z2 & u(4294967288)) << u(4)) & M32) ^ ((((z2 << u(2)) & M32) ^ z2) >> u(25))
        z3 = (((z3 & u(4294967280)) << u(17)) & M32) ^ ((((z3 << u(3)) & M32) ^ z3) >> u(11))
        z4 = (z4 * u(1664525) + u(1013904223)) & M32
        return z1, z2, z3, z4

    z1, z2, z3, z4 = step(z1, z2, z3, z4)
    out = np.empty(n, dtype=np.uint32)
    rows = (n + 65535) // 65536
    for r in range(rows):
        t = np.full(nt, 0xFFFFFFFF, dtype=np.uint64)
        for _ in range(and_count + 1):
            z1, z2, z3, z4 = step(z1, z2, z3, z4)
            t &= z1 ^ z2 ^ z3 ^ z4
        lo = r * 65536
        hi = min(n, lo + 65536)
        out[lo:hi] = t[: hi - lo].astype(np.uint32)
    return out


def to_bits(u, key_type):
    u = u.astype(np.uint32)
    if key_type == 1:
        return u ^ np.uint32(0x80000000)
    if key_type == 2:
        neg = (u >> np.uint32(31)).astype(bool)
        return np.where(neg, ~u, u | np.uint32(0x80000000)).astype(np.uint32)
    return u


def sort_np(keys, key_type, order, vals=None):
    perm = np.argsort(to_bits(keys, key_type), kind="stable")
    if order == 1:
        perm = perm[::-1]
    return (keys[perm], None if vals is None else vals[perm])


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


# (n, seed, and_count, key_type, order, value_bytes)
CASES = [
    (1, 1, 0, 0, 0, 0), (2, 5, 0, 0, 1, 0), (64, 3, 0, 0, 0, 4), (65, 3, 0, 1, 0, 4), (1000, 7, 0, 0, 0, 0),
    (1000, 7, 1, 2, 1, 4), (2048, 11, 4, 0, 0, 8), (7680, 7680, 0, 0, 0, 0), (7681, 7681, 0, 0, 0, 4),
    (8192, 8192, 0, 0, 0, 0), (8193, 8193, 0, 1, 1, 0), (15360, 15360, 0, 2, 0, 4), (16385, 16385, 0, 0, 0, 8),
    (65536, 10, 0, 0, 0, 0), (65536, 10, 2, 2, 1, 4), (100003, 10, 0, 0, 0, 0), (1 << 20, 10, 0, 0, 0, 0),
    (1 << 20, 11, 3, 0, 0, 8), ((1 << 20) + 3, 12, 0, 1, 1, 4),
]


def main():
    out = {"cases": np.array(CASES, dtype=np.int64)}
    for ci, (n, seed, andc, kt, order, vb) in enumerate(CASES):
        keys = init_random_np(n, seed, andc)
        vals = None
        if vb:
            # value = original index: exposes stability (the reference uses value = key)
            vals = np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        sk, sv = sort_np(keys, kt, order, vals)
        out[f"c{ci}_in_crc"] = np.uint32(crc(keys))
        out[f"c{ci}_out_crc"] = np.uint32(crc(sk))
        out[f"c{ci}_in_head"] = keys[:16].copy()
        out[f"c{ci}_out_head"] = sk[:16].copy()
        out[f"c{ci}_out_tail"] = sk[-16:].copy()
        if vb:
            out[f"c{ci}_vout_crc"] = np.uint32(crc(sv))
            out[f"c{ci}_vout_head"] = sv[:16].copy()
        if n <= 2048:
            out[f"c{ci}_in"] = keys
            out[f"c{ci}_out"] = sk
            if vb:
                out[f"c{ci}_vout"] = sv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "onesweep_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
