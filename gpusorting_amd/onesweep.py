"""Host-side mirror of the reference's OneSweep dispatch surface, over the C-ABI.

Reference interfaces mirrored (paths relative to the reference root):
  * ``GPUSortingConfig`` / MODE / ORDER / KEY_TYPE / ENTROPY_PRESET enums —
    GPUSortingD3D12/GPUSorting.h:40-86
  * ``OneSweep`` — D3D12 ``OneSweep(device, deviceInfo, ORDER, KEY_TYPE[, PAYLOAD_TYPE])``
    (GPUSortingD3D12/OneSweep.h:16-27) with Unity's caller-owned-buffer
    ``Sort(...)`` (GPUSortingUnity/Runtime/OneSweep.cs:297-427)
  * ``OneSweepDispatcher`` — the CUDA tree's class, same method names, arguments
    and print format (GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:17-392)

PyTorch is used only for device memory and the current HIP stream; every
computation goes through libgpusort.so (hand-written gfx950 kernels).  There is
no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import check

MODE_KEYS_ONLY, MODE_PAIRS = 0, 1
ORDER_ASCENDING, ORDER_DESCENDING = 0, 1
KEY_UINT32, KEY_INT32, KEY_FLOAT32 = 0, 1, 2
KEY_UINT64, KEY_INT64, KEY_FLOAT64 = 3, 4, 5  # 8-byte keys: eight passes planned by one histogram sweep
PAYLOAD_UINT32, PAYLOAD_INT32, PAYLOAD_FLOAT32 = 0, 1, 2
ENTROPY_PRESET_1, ENTROPY_PRESET_2, ENTROPY_PRESET_3, ENTROPY_PRESET_4, ENTROPY_PRESET_5 = range(5)
_ENT_LOOKUP = (1.0, 0.811, 0.544, 0.337, 0.201)  # OneSweepDispatcher.cuh:201

_KEY_DTYPES = {torch.uint32: KEY_UINT32, torch.int32: KEY_INT32, torch.float32: KEY_FLOAT32}


@dataclass
class GPUSortingConfig:
    """GPUSortingD3D12/GPUSorting.h:70-76."""
    sortingMode: int = MODE_KEYS_ONLY
    sortingOrder: int = ORDER_ASCENDING
    sortingKeyType: int = KEY_UINT32
    sortingPayloadType: int = PAYLOAD_UINT32


def _stream_ptr(stream=None) -> int:
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous tensor on the GPU")


def _require_room(t: torch.Tensor | None, n: int, name: str) -> None:
    """The C-ABI takes raw pointers: every buffer handed over must hold at least n elements."""
    if t is not None:
        _require_cuda(t, name)
        if n > t.numel():
            raise ValueError(f"{name} holds {t.numel()} elements, n = {n}")


def init_random(keys: torch.Tensor, seed: int, entropy_preset: int = ENTROPY_PRESET_1,
                values: torch.Tensor | None = None, n: int | None = None) -> None:
    """InitRandom<<<256,256>>> (GPUSortingCUDA/UtilityKernels.cuh:53-117): fills keys (and values = key)."""
    _require_cuda(keys, "keys")
    n = keys.numel() if n is None else n
    vb = 0 if values is None else values.element_size()
    check(_lib.load().gs_init_random(keys.data_ptr(), None if values is None else values.data_ptr(), vb,
                                     int(entropy_preset), seed & 0xFFFFFFFF, n, _stream_ptr()), "gs_init_random")


def validate(keys: torch.Tensor, values: torch.Tensor | None = None, n: int | None = None,
             key_type: int = KEY_UINT32, order: int = ORDER_ASCENDING) -> int:
    """Validate (GPUSortingCUDA/UtilityKernels.cuh:402-479): number of adjacent inversions (0 == sorted)."""
    _require_cuda(keys, "keys")
    n = keys.numel() if n is None else n
    err = C.c_uint32(0xFFFFFFFF)
    vb = 0 if values is None else values.element_size()
    check(_lib.load().gs_validate(keys.data_ptr(), None if values is None else values.data_ptr(), vb, n, key_type,
                                  order, C.byref(err), _stream_ptr()), "gs_validate")
    return int(err.value)


class OneSweep:
    """One sorter object == one ``gs_onesweep`` handle (scan state) + lazily sized alt buffers."""

    def __init__(self, max_keys: int, order: int = ORDER_ASCENDING, key_type: int = KEY_UINT32,
                 mode: int = MODE_KEYS_ONLY, value_bytes: int = 0, device: int | None = None, **options):
        """``options``: fields of ``gs_onesweep_options`` (include/gpusort.h), e.g. ``mid_path=0``, ``plan=1``; what is not given
        comes from the GPUSORT_* environment variables of the test / tuning harness (``_lib.onesweep_options_from_env``), then
        from the library's defaults.  The library itself reads no environment."""
        if not torch.cuda.is_available():
            raise RuntimeError("gpusorting_amd needs a GPU: the product path has no CPU fallback")
        self._lib = _lib.load()
        if device is not None:
            torch.cuda.set_device(device)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.max_keys = int(max_keys)
        self.order, self.key_type, self.mode = order, key_type, mode
        self.value_bytes = value_bytes if mode == MODE_PAIRS else 0
        if mode == MODE_PAIRS and self.value_bytes == 0:
            self.value_bytes = 4
        h = C.c_void_p()
        if hasattr(self._lib, "gs_onesweep_create_ex"):
            opts = _lib.onesweep_options_from_env(**options)
            check(self._lib.gs_onesweep_create_ex(C.byref(h), self.max_keys, mode, self.value_bytes, C.byref(opts)), "gs_onesweep_create_ex")
        else:  # (A/B runs against a build from before the options struct: GPUSORT_LIB=...)
            check(self._lib.gs_onesweep_create(C.byref(h), self.max_keys, mode, self.value_bytes), "gs_onesweep_create")
        self._h = h
        self._alt_keys = None
        self._alt_vals = None

    @classmethod
    def _borrow(cls, handle, max_keys: int, mode: int, value_bytes: int, key_type: int = KEY_UINT32) -> "OneSweep":
        """A view of a gs_onesweep handle somebody else owns (the local engine inside a gs_mgpu context)."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.max_keys, self.order, self.key_type, self.mode = int(max_keys), ORDER_ASCENDING, key_type, mode
        self.value_bytes = value_bytes if mode == MODE_PAIRS else 0
        self._h = C.c_void_p(handle)
        self._borrowed = True
        self._alt_keys = self._alt_vals = None
        return self

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._lib.gs_onesweep_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_config(cls, max_keys: int, cfg: GPUSortingConfig, value_bytes: int = 4) -> "OneSweep":
        return cls(max_keys, cfg.sortingOrder, cfg.sortingKeyType, cfg.sortingMode,
                   value_bytes if cfg.sortingMode == MODE_PAIRS else 0)

    # -- helpers ------------------------------------------------------------
    @property
    def partition_size(self) -> int:
        return int(self._lib.gs_onesweep_get_partition_size(self._h))

    def set_shape(self, threads: int, keys_per_thread: int) -> None:
        check(self._lib.gs_onesweep_set_shape(self._h, threads, keys_per_thread), "gs_onesweep_set_shape")

    @property
    def rank_mode(self) -> int:
        """0 = ballot multi-split, 1 = returning LDS atomic (the library's choice after its device probe, or the caller's)."""
        return int(self._lib.gs_onesweep_get_rank_mode(self._h))

    def set_rank_mode(self, mode: int) -> None:
        check(self._lib.gs_onesweep_set_rank_mode(self._h, mode), "gs_onesweep_set_rank_mode")

    def set_small_path(self, on: bool) -> None:
        check(self._lib.gs_onesweep_set_small_path(self._h, 1 if on else 0), "gs_onesweep_set_small_path")

    def set_mid_path(self, on: bool) -> None:
        """Two-launch MSD + bucket sort for single-tile limit < n <= 2^20 (keys-only: 2^23, 4-byte values: 2^22; default on)."""
        check(self._lib.gs_onesweep_set_mid_path(self._h, 1 if on else 0), "gs_onesweep_set_mid_path")

    def set_skip_passes(self, on: bool) -> None:
        """Identity passes (one digit value for all keys) are dropped in pairs on the device (default on)."""
        check(self._lib.gs_onesweep_set_skip_passes(self._h, 1 if on else 0), "gs_onesweep_set_skip_passes")

    def set_plan(self, plan) -> None:
        """Large sorts of 32-bit keys, keys-only and pairs (4- / 8-byte values): 0 the library picks (two-level plan from 3 x 2^24 keys /
        2^25 + 1 pairs up when the device finds the keys near-uniform, LSD passes otherwise), 1 the four LSD passes only, 2 the
        two-level plan wherever it can run (tests)."""
        check(self._lib.gs_onesweep_set_plan(self._h, int(plan)), "gs_onesweep_set_plan")

    def last_plan(self) -> dict:
        """What the device decided for the last sort (synchronises): {'two_level': bool, 'largest_bucket': int}."""
        p, b = C.c_uint32(0), C.c_uint32(0)
        check(self._lib.gs_onesweep_last_plan(self._h, C.byref(p), C.byref(b), _stream_ptr()), "gs_onesweep_last_plan")
        return {"two_level": bool(p.value), "largest_bucket": int(b.value)}

    @property
    def key_bytes(self) -> int:
        return 8 if self.key_type >= KEY_UINT64 else 4

    def _alts(self, n: int, values: torch.Tensor | None):
        if self._alt_keys is None or self._alt_keys.numel() < n or self._alt_keys.element_size() != self.key_bytes:
            self._alt_keys = torch.empty(max(n, 1), dtype=torch.int64 if self.key_bytes == 8 else torch.int32, device=self.device)
        if values is not None and (self._alt_vals is None or self._alt_vals.numel() < n
                                   or self._alt_vals.element_size() != values.element_size()):
            dt = torch.int32 if values.element_size() == 4 else torch.int64
            self._alt_vals = torch.empty(max(n, 1), dtype=dt, device=self.device)
        return self._alt_keys, (self._alt_vals if values is not None else None)

    # -- the hot path ---------------------------------------------------------
    def sort(self, keys: torch.Tensor, values: torch.Tensor | None = None, n: int | None = None,
             alt_keys: torch.Tensor | None = None, alt_values: torch.Tensor | None = None, stream=None) -> None:
        """Sort ``keys[:n]`` (and ``values[:n]``) in place on the current stream (asynchronous).

        Unity ``OneSweep.Sort`` (OneSweep.cs:297-427): the caller may pass its own
        temp buffers (``alt_*``); otherwise the object keeps a pair.
        """
        _require_cuda(keys, "keys")
        n = keys.numel() if n is None else int(n)
        if keys.element_size() != self.key_bytes:
            raise ValueError(f"keys must be a {8 * self.key_bytes}-bit type for this sorter's key type")
        if (values is not None) != (self.mode == MODE_PAIRS):
            raise ValueError("values must be given exactly when the sorter was built with MODE_PAIRS")
        _require_room(keys, n, "keys")
        _require_room(values, n, "values")
        if alt_keys is None or (values is not None and alt_values is None):
            own_alt_keys, own_alt_vals = self._alts(n, values)
            if alt_keys is None:
                alt_keys = own_alt_keys
            if alt_values is None:
                alt_values = own_alt_vals
        _require_room(alt_keys, n, "alt_keys")
        _require_room(alt_values if values is not None else None, n, "alt_values")
        if alt_keys.element_size() != self.key_bytes or (values is not None and alt_values.element_size() != values.element_size()):
            raise ValueError("alt buffers must have the element size of the buffers they shadow")
        s = _stream_ptr(stream)
        if values is None:
            st = self._lib.gs_onesweep_sort_keys(self._h, keys.data_ptr(), alt_keys.data_ptr(), n, self.key_type,
                                                 self.order, s)
        else:
            _require_cuda(values, "values")
            if values.element_size() != self.value_bytes:
                raise ValueError(f"values must be {self.value_bytes}-byte elements")
            st = self._lib.gs_onesweep_sort_pairs(self._h, keys.data_ptr(), values.data_ptr(), alt_keys.data_ptr(),
                                                  alt_values.data_ptr(), n, self.key_type, self.order, s)
        check(st, "gs_onesweep_sort")

    def check(self, stream=None) -> None:
        """Synchronise and raise if the device reported a look-back timeout."""
        check(self._lib.gs_onesweep_check(self._h, _stream_ptr(stream)), "gs_onesweep_check")

    def check_state(self, stream=None) -> dict:
        """Post-call invariants of the chained-scan state (gs_debug_check_state): synchronises and returns the report."""
        rep = (C.c_uint64 * 8)()
        check(self._lib.gs_debug_check_state(self._h, rep, _stream_ptr(stream)), "gs_debug_check_state")
        return {"rows_not_inclusive": int(rep[0]), "rows_not_monotone": int(rep[1]), "chains_short_of_tickets": int(rep[2]),
                "hist_words_nonzero": int(rep[3]), "keys_per_pass": [int(rep[4 + q]) for q in range(4)]}

    # -- structural entry points ------------------------------------------------
    def global_histogram(self, keys: torch.Tensor, n: int | None = None) -> np.ndarray:
        n = keys.numel() if n is None else int(n)
        _require_room(keys, n, "keys")
        out = (C.c_uint32 * 1024)()
        check(self._lib.gs_onesweep_global_histogram(self._h, keys.data_ptr(), n, self.key_type, out, _stream_ptr()),
              "gs_onesweep_global_histogram")
        return np.frombuffer(out, dtype=np.uint32).reshape(4, 256).copy()

    def scan_rows(self, keys: torch.Tensor, n: int | None = None) -> np.ndarray:
        """GlobalHistogram + Scan; the raw first descriptor row of each of the four passes: (digit start << 2) | 2."""
        n = keys.numel() if n is None else int(n)
        _require_room(keys, n, "keys")
        out = (C.c_uint32 * 1024)()
        check(self._lib.gs_onesweep_scan(self._h, keys.data_ptr(), n, self.key_type, out, _stream_ptr()), "gs_onesweep_scan")
        return np.frombuffer(out, dtype=np.uint32).reshape(4, 256).copy()

    def digit_pass(self, keys_in: torch.Tensor, keys_out: torch.Tensor, pass_index: int, n: int | None = None,
                   values_in: torch.Tensor | None = None, values_out: torch.Tensor | None = None,
                   reverse_index: bool = False) -> None:
        n = keys_in.numel() if n is None else int(n)
        for t, name in ((keys_in, "keys_in"), (keys_out, "keys_out"), (values_in, "values_in"), (values_out, "values_out")):
            _require_room(t, n, name)
        check(self._lib.gs_onesweep_digit_pass(
            self._h, keys_in.data_ptr(), keys_out.data_ptr(),
            None if values_in is None else values_in.data_ptr(),
            None if values_out is None else values_out.data_ptr(), n, pass_index, self.key_type,
            1 if reverse_index else 0, _stream_ptr()), "gs_onesweep_digit_pass")

    def msd_prepare(self, keys: torch.Tensor, n: int | None = None) -> np.ndarray:
        """Top-byte histogram of ``keys[:n]`` (256 counts); leaves histogram + scan in the handle for msd_partition."""
        n = keys.numel() if n is None else int(n)
        _require_room(keys, n, "keys")
        out = (C.c_uint32 * 256)()
        check(self._lib.gs_onesweep_msd_prepare(self._h, keys.data_ptr(), n, self.key_type, out, _stream_ptr()),
              "gs_onesweep_msd_prepare")
        return np.frombuffer(out, dtype=np.uint32).copy()

    def msd_fine_histogram(self, keys: torch.Tensor, n: int | None = None) -> np.ndarray:
        """4096-bin histogram of the 12-bit key prefix (top byte, top nibble of the next byte); synchronous."""
        n = keys.numel() if n is None else int(n)
        _require_room(keys, n, "keys")
        out = (C.c_uint32 * 4096)()
        check(self._lib.gs_onesweep_msd_fine_histogram(self._h, keys.data_ptr(), n, self.key_type, out, _stream_ptr()),
              "gs_onesweep_msd_fine_histogram")
        return np.frombuffer(out, dtype=np.uint32).copy()

    def msd_partition(self, keys_in: torch.Tensor, keys_out: torch.Tensor, n: int | None = None,
                      values_in: torch.Tensor | None = None, values_out: torch.Tensor | None = None) -> None:
        n = keys_in.numel() if n is None else int(n)
        for t, name in ((keys_in, "keys_in"), (keys_out, "keys_out"), (values_in, "values_in"), (values_out, "values_out")):
            _require_room(t, n, name)
        check(self._lib.gs_onesweep_msd_partition(
            self._h, keys_in.data_ptr(), keys_out.data_ptr(),
            None if values_in is None else values_in.data_ptr(),
            None if values_out is None else values_out.data_ptr(), n, _stream_ptr()), "gs_onesweep_msd_partition")

    # -- profiling ------------------------------------------------------------------
    def set_profiling(self, enabled: bool) -> None:
        check(self._lib.gs_onesweep_set_profiling(self._h, 1 if enabled else 0), "gs_onesweep_set_profiling")

    def get_profile(self) -> dict:
        ms = (C.c_float * _lib.GS_PROFILE_SLOTS)()
        check(self._lib.gs_onesweep_get_profile(self._h, ms), "gs_onesweep_get_profile")
        names = ("clear", "global_histogram", "scan", "pass0", "pass1", "pass2", "pass3", "total")
        return {k: float(v) for k, v in zip(names, ms)}


class OneSweepDispatcher:
    """The CUDA tree's benchmark/test class, method for method.

    GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:17-392.  Like the reference it
    owns its buffers (``m_sort``, ``m_alt``, payloads) and generates inputs on
    the device; prints the same lines.  ``quick`` shortens ``TestAll*`` for CI
    (the reference has no such switch).
    """

    def __init__(self, keysOnly: bool, maxSize: int, value_bytes: int = 4, out=print):
        self.k_keysOnly = bool(keysOnly)
        self.k_maxSize = int(maxSize)
        self._out = out
        self._sorter = OneSweep(self.k_maxSize, mode=MODE_KEYS_ONLY if keysOnly else MODE_PAIRS,
                                value_bytes=0 if keysOnly else value_bytes)
        self.k_partitionSize = self._sorter.partition_size
        dev = self._sorter.device
        self.m_sort = torch.empty(self.k_maxSize, dtype=torch.int32, device=dev)
        self.m_alt = torch.empty(self.k_maxSize, dtype=torch.int32, device=dev)
        self.m_sortPayload = self.m_altPayload = None
        if not keysOnly:
            dt = torch.int32 if value_bytes == 4 else torch.int64
            self.m_sortPayload = torch.empty(self.k_maxSize, dtype=dt, device=dev)
            self.m_altPayload = torch.empty(self.k_maxSize, dtype=dt, device=dev)

    # private members of the reference, kept with their names
    def DispatchKernelsKeysOnly(self, size: int) -> None:
        self._sorter.sort(self.m_sort, n=size, alt_keys=self.m_alt)

    def DispatchKernelsPairs(self, size: int) -> None:
        self._sorter.sort(self.m_sort, self.m_sortPayload, n=size, alt_keys=self.m_alt, alt_values=self.m_altPayload)

    def DispatchValidateKeys(self, size: int) -> bool:
        return validate(self.m_sort, n=size) == 0

    def DispatchValidatePairs(self, size: int) -> bool:
        return validate(self.m_sort, self.m_sortPayload, n=size) == 0

    def _test_all(self, pairs: bool, quick: bool) -> bool:
        if self.k_maxSize < (1 << 28) and not quick:
            self._out(f"This test requires a minimum initialized size of {1 << 28}. "
                      f"Reinitialize the object to at least {1 << 28}.")
            return False
        if pairs and self.k_keysOnly:
            self._out("Error, object was intialized for keys only")
            return False
        self._out(f"Beginning GPUSorting OneSweep {'pairs' if pairs else 'keys'} validation test: ")
        P = self.k_partitionSize
        sizes = list(range(P, 2 * P + 1, 97 if quick else 1))
        if quick and sizes[-1] != 2 * P:
            sizes.append(2 * P)
        big = [e for e in (26, 27, 28) if (1 << e) <= self.k_maxSize]
        passed, dots = 0, []
        for i in sizes:
            init_random(self.m_sort, i, ENTROPY_PRESET_1, self.m_sortPayload if pairs else None, n=i)
            (self.DispatchKernelsPairs if pairs else self.DispatchKernelsKeysOnly)(i)
            if (self.DispatchValidatePairs if pairs else self.DispatchValidateKeys)(i):
                passed += 1
            else:
                self._out(f"\n Test failed at size {i} ")
            if not (i & 255):
                dots.append(".")
        self._out("".join(dots))
        for e in big:
            init_random(self.m_sort, e, ENTROPY_PRESET_1, self.m_sortPayload if pairs else None, n=1 << e)
            (self.DispatchKernelsPairs if pairs else self.DispatchKernelsKeysOnly)(1 << e)
            if (self.DispatchValidatePairs if pairs else self.DispatchValidateKeys)(1 << e):
                passed += 1
            else:
                self._out(f"\n Test failed at size {1 << e} ")
        self._sorter.check()
        expected = len(sizes) + len(big)
        if passed == expected:
            self._out(f"{passed}/{passed} All tests passed.\n")
        else:
            self._out(f"{passed}/{expected} Test failed.\n")
        return passed == expected

    def TestAllKeysOnly(self, quick: bool = False) -> bool:
        return self._test_all(False, quick)

    def TestAllPairs(self, quick: bool = False) -> bool:
        return self._test_all(True, quick)

    def _batch_timing(self, pairs: bool, size: int, batchCount: int, seed: int, entropyPreset: int) -> float:
        if pairs and self.k_keysOnly:
            self._out("Error, object was intialized for keys only")
            return 0.0
        if size > self.k_maxSize:
            self._out("Error, requested test size exceeds max initialized size. ")
            return 0.0
        self._out(f"Beginning GPUSorting OneSweep {'pairs' if pairs else 'keys'} batch timing test at:")
        self._out(f"Size: {size}")
        self._out(f"Entropy: {_ENT_LOOKUP[entropyPreset]:f} bits")
        self._out(f"Test size: {batchCount}")
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        total = 0.0
        for i in range(batchCount + 1):
            # the reference's BatchTimingPairs leaves the payload uninitialised
            # (OneSweepDispatcher.cuh:269-273); we initialise value = key.
            init_random(self.m_sort, i + seed, entropyPreset, self.m_sortPayload if pairs else None, n=size)
            torch.cuda.synchronize()
            start.record()
            (self.DispatchKernelsPairs if pairs else self.DispatchKernelsKeysOnly)(size)
            stop.record()
            stop.synchronize()
            if i:
                total += start.elapsed_time(stop)
        total /= 1000.0
        self._out(f"Total time elapsed: {total:f}")
        rate = size / total * batchCount if total > 0 else 0.0
        self._out(f"Estimated speed at {size} 32-bit elements: {rate:E} keys/sec\n")
        return rate

    def BatchTimingKeysOnly(self, size: int, batchCount: int, seed: int, entropyPreset: int) -> float:
        return self._batch_timing(False, size, batchCount, seed, entropyPreset)

    def BatchTimingPairs(self, size: int, batchCount: int, seed: int, entropyPreset: int) -> float:
        return self._batch_timing(True, size, batchCount, seed, entropyPreset)
