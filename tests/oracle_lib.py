"""numpy-facing wrapper of oracle/libgs_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  It builds the oracle with gcc on first use if the .so is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "libgs_oracle.so")

KEY_U32, KEY_I32, KEY_F32 = 0, 1, 2
ASC, DESC = 0, 1

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        v, u32, i = C.c_void_p, C.c_uint32, C.c_int
        lib.gso_init_random.argtypes = [v, v, u32, u32, u32, u32]
        lib.gso_init_random.restype = None
        lib.gso_key_to_bits.argtypes = [u32, i]
        lib.gso_key_to_bits.restype = u32
        lib.gso_bits_to_key.argtypes = [u32, i]
        lib.gso_bits_to_key.restype = u32
        lib.gso_global_histogram.argtypes = [v, u32, i, v]
        lib.gso_global_histogram.restype = None
        lib.gso_scan.argtypes = [v, v]
        lib.gso_scan.restype = None
        lib.gso_digit_binning_pass.argtypes = [v, v, v, v, u32, u32, u32, i, i]
        lib.gso_digit_binning_pass.restype = None
        lib.gso_onesweep_sort.argtypes = [v, v, v, v, u32, u32, i, i]
        lib.gso_onesweep_sort.restype = None
        lib.gso_std_sort.argtypes = [v, v, u32, u32, i, i]
        lib.gso_std_sort.restype = None
        lib.gso_std_sort_parallel.argtypes = [v, u32, u32]
        lib.gso_std_sort_parallel.restype = None
        lib.gso_sort_permutation_parallel.argtypes = [v, u32, i, i, u32, v]
        lib.gso_sort_permutation_parallel.restype = None
        lib.gso_key64_to_bits.argtypes = [C.c_uint64, i]
        lib.gso_key64_to_bits.restype = C.c_uint64
        lib.gso_std_sort64.argtypes = [v, v, u32, u32, i, i]
        lib.gso_std_sort64.restype = None
        lib.gso_digit_binning_pass64.argtypes = [v, v, v, v, u32, u32, u32, i, i]
        lib.gso_digit_binning_pass64.restype = None
        lib.gso_validate.argtypes = [v, v, u32, u32, i, i]
        lib.gso_validate.restype = u32
        lib.gso_msd_splitters.argtypes = [v, u32, v]
        lib.gso_msd_splitters.restype = None
        lib.gso_hardware_threads.restype = C.c_uint

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    @staticmethod
    def _vb(vals):
        return 0 if vals is None else vals.dtype.itemsize

    def init_random(self, n, seed, and_count=0, value_bytes=0):
        keys = np.empty(n, dtype=np.uint32)
        vals = None
        if value_bytes:
            vals = np.empty(n, dtype=np.uint32 if value_bytes == 4 else np.uint64)
        self.lib.gso_init_random(self._p(keys), self._p(vals), value_bytes, and_count, seed & 0xFFFFFFFF, n)
        return (keys, vals) if value_bytes else keys

    def global_histogram(self, keys, key_type=KEY_U32):
        h = np.zeros(1024, dtype=np.uint32)
        self.lib.gso_global_histogram(self._p(keys), keys.size, key_type, self._p(h))
        return h.reshape(4, 256)

    def scan(self, hist):
        out = np.zeros(1024, dtype=np.uint32)
        self.lib.gso_scan(self._p(np.ascontiguousarray(hist.reshape(-1))), self._p(out))
        return out.reshape(4, 256)

    def digit_pass(self, keys, shift, key_type=KEY_U32, vals=None, reverse=False):
        ko = np.empty_like(keys)
        vo = None if vals is None else np.empty_like(vals)
        self.lib.gso_digit_binning_pass(self._p(keys), self._p(ko), self._p(vals), self._p(vo), self._vb(vals),
                                        keys.size, shift, key_type, 1 if reverse else 0)
        return ko if vals is None else (ko, vo)

    def onesweep_sort(self, keys, key_type=KEY_U32, order=ASC, vals=None):
        k = keys.copy()
        ak = np.empty_like(k)
        v = None if vals is None else vals.copy()
        av = None if vals is None else np.empty_like(v)
        self.lib.gso_onesweep_sort(self._p(k), self._p(ak), self._p(v), self._p(av), self._vb(vals), k.size, key_type, order)
        return k if vals is None else (k, v)

    def std_sort(self, keys, key_type=KEY_U32, order=ASC, vals=None):
        k = keys.copy()
        v = None if vals is None else vals.copy()
        self.lib.gso_std_sort(self._p(k), self._p(v), self._vb(vals), k.size, key_type, order)
        return k if vals is None else (k, v)

    def std_sort64(self, keys, key_type=KEY_U32, order=ASC, vals=None):
        """64-bit keys (uint64 array of native bit patterns); key_type 0/1/2 = uint64 / int64 / float64."""
        k = keys.copy()
        v = None if vals is None else vals.copy()
        self.lib.gso_std_sort64(self._p(k), self._p(v), self._vb(vals), k.size, key_type, order)
        return k if vals is None else (k, v)

    def digit_pass64(self, keys, shift, key_type=KEY_U32, vals=None, reverse=False):
        ko = np.empty_like(keys)
        vo = None if vals is None else np.empty_like(vals)
        self.lib.gso_digit_binning_pass64(self._p(keys), self._p(ko), self._p(vals), self._p(vo), self._vb(vals),
                                          keys.size, shift, key_type, 1 if reverse else 0)
        return ko if vals is None else (ko, vo)

    def std_sort_parallel(self, keys, threads):
        k = keys.copy()
        self.lib.gso_std_sort_parallel(self._p(k), k.size, threads)
        return k

    def sort_permutation_parallel(self, keys, key_type=KEY_U32, order=ASC, threads=None):
        """perm[j] = original index of the element at sorted position j (stable by key; descending = reverse)."""
        perm = np.empty(keys.size, dtype=np.uint32)
        self.lib.gso_sort_permutation_parallel(self._p(keys), keys.size, key_type, order,
                                               threads or self.hardware_threads(), self._p(perm))
        return perm

    def validate(self, keys, key_type=KEY_U32, order=ASC, vals=None):
        return int(self.lib.gso_validate(self._p(keys), self._p(vals), self._vb(vals), keys.size, key_type, order))

    def msd_splitters(self, hist256, world):
        h = np.ascontiguousarray(hist256, dtype=np.uint64)
        fb = np.zeros(world + 1, dtype=np.uint32)
        self.lib.gso_msd_splitters(self._p(h), world, self._p(fb))
        return fb

    def hardware_threads(self):
        return int(self.lib.gso_hardware_threads())


def build(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ("gs_oracle.cpp", "gs_oracle.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return SO


_cached = None


def load():
    global _cached
    if _cached is None:
        _cached = Oracle(C.CDLL(build()))
    return _cached
