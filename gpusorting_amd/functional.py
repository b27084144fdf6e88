"""Tensor-in / tensor-out convenience on top of ``OneSweep`` (PyTorch-ROCm plumbing only: dtype -> key type,
handle cache, buffers).  The reference has no such layer; it is what a torch user types:

    sorted_keys = gpusorting_amd.sort(keys)                                  # int32 / uint32-as-int32 / float32
    sorted_keys, sorted_vals = gpusorting_amd.sort(keys, values, descending=True)
    gpusorting_amd.sort_(keys, values)                                       # in place

Semantics are the library's: stable LSD radix sort; descending = exact reverse of the stable ascending result;
float keys ordered by the order-preserving bit flip (-0 < +0, NaNs by bit pattern); values bit-copied.
"""
from __future__ import annotations

import torch

from .onesweep import KEY_FLOAT32, KEY_INT32, KEY_UINT32, MODE_KEYS_ONLY, MODE_PAIRS, ORDER_ASCENDING, ORDER_DESCENDING, OneSweep

_KEY_TYPE = {torch.int32: KEY_INT32, torch.float32: KEY_FLOAT32, torch.uint32: KEY_UINT32}
_cache: dict = {}


def _sorter(device: torch.device, n: int, key_type: int, order: int, value_bytes: int) -> OneSweep:
    """One cached handle per (device, stream, type, order, value width); re-created when the size outgrows it."""
    key = (device.index, int(torch.cuda.current_stream(device).cuda_stream), key_type, order, value_bytes)
    s = _cache.get(key)
    if s is None or s.max_keys < n:
        if s is not None:
            s.close()
        cap = 1 << max(int(n - 1).bit_length(), 16)  # next power of two: few re-creations while sizes wander
        s = OneSweep(min(cap, (1 << 30) - 1), order, key_type, MODE_PAIRS if value_bytes else MODE_KEYS_ONLY, value_bytes,
                     device=device.index)
        _cache[key] = s
    return s


def sort_(keys: torch.Tensor, values: torch.Tensor | None = None, descending: bool = False, unsigned: bool = False) -> None:
    """Sort ``keys`` (and carry ``values``) in place on the current stream.  ``unsigned=True`` treats int32 storage
    as uint32 keys (torch has little uint32 support)."""
    if keys.dim() != 1 or not keys.is_contiguous() or keys.device.type != "cuda":
        raise ValueError("keys must be a contiguous 1-D device tensor")
    if keys.dtype not in _KEY_TYPE:
        raise TypeError(f"unsupported key dtype {keys.dtype}: 32-bit keys only (int32, uint32, float32)")
    kt = KEY_UINT32 if (unsigned and keys.dtype == torch.int32) else _KEY_TYPE[keys.dtype]
    vb = 0
    if values is not None:
        if values.shape != keys.shape or not values.is_contiguous() or values.device != keys.device:
            raise ValueError("values must match keys in shape and device and be contiguous")
        vb = values.element_size()
        if vb not in (4, 8):
            raise TypeError("values must be 4 or 8 bytes wide")
    n = keys.numel()
    if n <= 1:
        return
    with torch.cuda.device(keys.device):
        s = _sorter(keys.device, n, kt, ORDER_DESCENDING if descending else ORDER_ASCENDING, vb)
        s.sort(keys.view(torch.int32) if keys.dtype != torch.int32 else keys, values, n=n)


def sort(keys: torch.Tensor, values: torch.Tensor | None = None, descending: bool = False, unsigned: bool = False):
    """Out-of-place: returns ``sorted_keys`` or ``(sorted_keys, sorted_values)``."""
    k = keys.clone()
    v = None if values is None else values.clone()
    sort_(k, v, descending, unsigned)
    return k if v is None else (k, v)


def argsort(keys: torch.Tensor, descending: bool = False, unsigned: bool = False) -> torch.Tensor:
    """Stable permutation that sorts ``keys`` (int32 indices; n < 2^30)."""
    idx = torch.arange(keys.numel(), dtype=torch.int32, device=keys.device)
    k = keys.clone()
    sort_(k, idx, descending, unsigned)
    return idx
