"""gpusorting_amd — MI355X-native OneSweep radix sort behind the reference's
GPUSortBase / OneSweep / OneSweepDispatcher dispatch surface.

Only the hot path of b0nes164/GPUSorting named by BASELINE.json is here:
  csrc/      hand-written gfx950 HIP kernels + the C-ABI (include/gpusort.h)
  onesweep   host-side mirror of the reference interface (ctypes over the C-ABI)
  sharded    one-process-per-GPU MSD split + RCCL all-to-all-v + local OneSweep
  functional sort / sort_ / argsort on torch tensors (plumbing over OneSweep)
"""
from .onesweep import (  # noqa: F401
    ENTROPY_PRESET_1, ENTROPY_PRESET_2, ENTROPY_PRESET_3, ENTROPY_PRESET_4, ENTROPY_PRESET_5,
    KEY_FLOAT32, KEY_FLOAT64, KEY_INT32, KEY_INT64, KEY_UINT32, KEY_UINT64, MODE_KEYS_ONLY, MODE_PAIRS, ORDER_ASCENDING, ORDER_DESCENDING,
    GPUSortingConfig, OneSweep, OneSweepDispatcher, init_random, validate,
)
from ._lib import GpuSortError  # noqa: F401
from .functional import argsort, sort, sort_  # noqa: F401
