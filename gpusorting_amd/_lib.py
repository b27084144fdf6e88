"""ctypes binding of libgpusort.so (the C-ABI of include/gpusort.h).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``make`` into ``gpusorting_amd/lib/libgpusort.so``.  There is NO fallback: if
the library is missing, importing the binding raises — the product path never
routes through the CPU oracle or any eager PyTorch op.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPUSORT_LIB") or os.path.join(_HERE, "lib", "libgpusort.so")  # env: ablation builds

GS_OK, GS_ERR_ARG, GS_ERR_SIZE, GS_ERR_HIP, GS_ERR_TIMEOUT, GS_ERR_MODE, GS_ERR_NO_DEVICE, GS_ERR_COMM = range(8)
GS_MGPU_UNIQUE_ID_BYTES = 128
GS_MAX_KEYS = (1 << 30) - 1
GS_PROFILE_SLOTS = 8

# every symbol include/gpusort.h declares: (name, restype, argtypes)
_u32, _vp, _int = C.c_uint32, C.c_void_p, C.c_int
_u32p, _u64p, _u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)

# gs_mgpu_transport (include/gpusort.h): the two functions a transport provides
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_size_t, _vp)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, _vp, _u32, C.POINTER(_vp), C.POINTER(_vp), _u32p, _u32p, _u32p, _u32p, _u32p, _vp)


class OneSweepOptions(C.Structure):
    """gs_onesweep_options (include/gpusort.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("shape_threads", C.c_uint32), ("shape_keys_per_thread", C.c_uint32),
                ("rank_mode", C.c_int32), ("small_path", C.c_int32), ("mid_path", C.c_int32), ("skip_passes", C.c_int32),
                ("position_chains", C.c_int32), ("position_chains_min_log2", C.c_uint32), ("key64_sweeps", C.c_int32),
                ("plan", C.c_int32), ("first_pass_big", C.c_int32), ("hist_blocks", C.c_uint32), ("debug_flags", C.c_uint32)]


class MgpuOptions(C.Structure):
    """gs_mgpu_options (include/gpusort.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("force_exchange", C.c_int32), ("overlap", C.c_int32), ("alltoallv", C.c_int32), ("by_bin", C.c_int32),
                ("sorter", OneSweepOptions)]


def onesweep_options_from_env(**overrides) -> "OneSweepOptions":
    """The library reads no environment variables; this harness does: GPUSORT_* names (tests, tools/, A/B runs) are
    translated into gs_onesweep_options here.  Keyword arguments win over the environment."""
    o = OneSweepOptions()
    load().gs_onesweep_options_default(C.byref(o))
    env = os.environ
    if "GPUSORT_SHAPE" in env:  # "512x16"
        try:
            t, k = env["GPUSORT_SHAPE"].lower().split("x")
            o.shape_threads, o.shape_keys_per_thread = int(t), int(k)
        except ValueError:
            pass
    for name, field in (("GPUSORT_RANK", "rank_mode"), ("GPUSORT_SMALL_PATH", "small_path"), ("GPUSORT_MID_PATH", "mid_path"),
                        ("GPUSORT_SKIP_PASSES", "skip_passes"), ("GPUSORT_POS", "position_chains"),
                        ("GPUSORT_POS_MIN_LOG2", "position_chains_min_log2"), ("GPUSORT_KEY64_SWEEPS", "key64_sweeps"),
                        ("GPUSORT_PLAN", "plan"), ("GPUSORT_FIRST_PASS_BIG", "first_pass_big"), ("GPUSORT_HIST_BLOCKS", "hist_blocks"),
                        ("GPUSORT_LS_EXP", "debug_flags"), ("GPUSORT_EXPMODE", "debug_flags")):
        if name in env:
            try:
                v = int(env[name], 0)
                setattr(o, field, (getattr(o, field) | v) if field == "debug_flags" else v)  # (two variables feed debug_flags: their bits add up)
            except ValueError:
                pass
    _apply_overrides(o, overrides)
    return o


def _apply_overrides(o, overrides) -> None:
    """Keyword overrides of an options struct: a name the struct does not have is a TypeError (setattr on a ctypes.Structure
    would silently create a Python attribute and the option would be ignored)."""
    known = {f[0] for f in o._fields_}
    for k, v in overrides.items():
        if k not in known:
            raise TypeError(f"{type(o).__name__} has no option {k!r} (known: {sorted(known)})")
        if v is not None:
            setattr(o, k, int(v))


def mgpu_options_from_env(**overrides) -> "MgpuOptions":
    o = MgpuOptions()
    load().gs_mgpu_options_default(C.byref(o))
    o.sorter = onesweep_options_from_env()  # the context's local sorter follows the same GPUSORT_* switches
    for name, field in (("GPUSORT_MGPU_FORCE_EXCHANGE", "force_exchange"), ("GPUSORT_MGPU_OVERLAP", "overlap"),
                        ("GPUSORT_MGPU_ALLTOALLV", "alltoallv"), ("GPUSORT_MGPU_BY_BIN", "by_bin")):
        if name in os.environ:
            try:
                setattr(o, field, int(os.environ[name], 0))
            except ValueError:
                pass
    _apply_overrides(o, overrides)
    return o


class MgpuTransport(C.Structure):
    _fields_ = [("user", _vp), ("all_gather_u32", ALL_GATHER_FN), ("exchange", EXCHANGE_FN)]


_PROTOS = [
    ("gs_version", C.c_char_p, []),
    ("gs_status_string", C.c_char_p, [_int]),
    ("gs_last_hip_error", _int, []),
    ("gs_onesweep_create", _int, [C.POINTER(_vp), _u32, _int, _u32]),
    ("gs_onesweep_options_default", None, [_vp]),
    ("gs_onesweep_create_ex", _int, [C.POINTER(_vp), _u32, _int, _u32, _vp]),
    ("gs_onesweep_destroy", _int, [_vp]),
    ("gs_onesweep_temp_bytes", C.c_size_t, [_u32]),
    ("gs_onesweep_partition_size", _u32, [_int, _u32]),
    ("gs_onesweep_sort_keys", _int, [_vp, _vp, _vp, _u32, _int, _int, _vp]),
    ("gs_onesweep_sort_pairs", _int, [_vp, _vp, _vp, _vp, _vp, _u32, _int, _int, _vp]),
    ("gs_onesweep_check", _int, [_vp, _vp]),
    ("gs_onesweep_set_shape", _int, [_vp, _u32, _u32]),
    ("gs_onesweep_get_partition_size", _u32, [_vp]),
    ("gs_onesweep_set_rank_mode", _int, [_vp, _int]),
    ("gs_onesweep_get_rank_mode", _int, [_vp]),
    ("gs_onesweep_set_small_path", _int, [_vp, _int]),
    ("gs_onesweep_set_skip_passes", _int, [_vp, _int]),
    ("gs_onesweep_set_mid_path", _int, [_vp, _int]),
    ("gs_onesweep_set_plan", _int, [_vp, _int]),
    ("gs_onesweep_last_plan", _int, [_vp, _u32p, _u32p, _vp]),
    ("gs_selftest_lds_atomic_order", _int, [_u32, _u32, C.POINTER(C.c_uint64), _vp]),
    ("gs_selftest_wave_primitives", _int, [_u32, _u32, _vp, _vp]),
    ("gs_debug_set_trace", _int, [_vp, _vp]),
    ("gs_debug_check_state", _int, [_vp, C.POINTER(C.c_uint64), _vp]),
    ("gs_debug_poke_status", _int, [_vp, _u32, _vp]),
    ("gs_debug_read_slab", _int, [_vp, _u32, _u32, _u32p, _vp]),
    ("gs_onesweep_global_histogram", _int, [_vp, _vp, _u32, _int, C.POINTER(_u32), _vp]),
    ("gs_onesweep_scan", _int, [_vp, _vp, _u32, _int, C.POINTER(_u32), _vp]),
    ("gs_onesweep_digit_pass", _int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _int, _int, _vp]),
    ("gs_onesweep_msd_prepare", _int, [_vp, _vp, _u32, _int, C.POINTER(_u32), _vp]),
    ("gs_onesweep_msd_partition", _int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("gs_onesweep_set_profiling", _int, [_vp, _int]),
    ("gs_onesweep_get_profile", _int, [_vp, C.POINTER(C.c_float)]),
    ("gs_init_random", _int, [_vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    ("gs_validate", _int, [_vp, _vp, _u32, _u32, _int, _int, C.POINTER(_u32), _vp]),
    ("gs_msd_splitters", _int, [C.POINTER(C.c_uint64), _u32, C.POINTER(_u32)]),
    ("gs_msd_splitters_n", _int, [C.POINTER(C.c_uint64), _u32, _u32, C.POINTER(_u32)]),
    ("gs_onesweep_msd_fine_histogram", _int, [_vp, _vp, _u32, _int, C.POINTER(_u32), _vp]),
    ("gs_mgpu_get_unique_id", _int, [_u8p]),
    ("gs_mgpu_create", _int, [C.POINTER(_vp), _u8p, _u32, _u32, _u32, _u32, _int, _u32]),
    ("gs_mgpu_options_default", None, [_vp]),
    ("gs_mgpu_create_ex", _int, [C.POINTER(_vp), _u8p, _u32, _u32, _u32, _u32, _int, _u32, _vp]),
    ("gs_mgpu_set_alltoallv", _int, [_vp, _int]),
    ("gs_mgpu_create_with_transport_ex", _int, [C.POINTER(_vp), C.POINTER(MgpuTransport), _u32, _u32, _u32, _u32, _int, _u32, _vp]),
    ("gs_mgpu_destroy", _int, [_vp]),
    ("gs_onesweep_sort_sharded", _int, [_vp, _vp, _vp, _u32, _int, _vp, _vp, _u32p, _vp]),
    ("gs_mgpu_get_profile", _int, [_vp, C.POINTER(C.c_float), _u64p, _u64p, _u32p]),
    ("gs_mgpu_last_plan", _int, [_vp, _u32p, _u32]),
    ("gs_mgpu_check", _int, [_vp, _vp]),
    ("gs_mgpu_debug_fail", _int, [_vp, _int]),
    ("gs_mgpu_sorter", _vp, [_vp]),
    ("gs_mgpu_set_force_exchange", _int, [_vp, _int]),
    ("gs_mgpu_last_layout", _int, [_vp, _u32p]),
    ("gs_msd_exchange_round", _int, [_u32p, _u32, _u32, _u32p, _int, _u32, _u32p, _u32p, _u32p, _u32p, _u32p]),
    ("gs_last_rccl_error", _int, []),
    ("gs_mgpu_create_with_transport", _int, [C.POINTER(_vp), C.POINTER(MgpuTransport), _u32, _u32, _u32, _u32, _int, _u32]),
    ("gs_msd_plan", _int, [_u32p, _u32, _u32, _u32, _u32, _u32p]),
    ("gs_debug_msd_plan_device", _int, [_u32p, _u32, _u32, _u32, _u32, _u32p, _vp]),
]
EXPORTED_SYMBOLS = [p[0] for p in _PROTOS]

_lib = None


class GpuSortError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = load().gs_status_string(status).decode()
        if status == GS_ERR_HIP:
            msg += f" (hipError {load().gs_last_hip_error()})"
        if status == GS_ERR_COMM:
            msg += f" (ncclResult {load().gs_last_rccl_error()})"
        super().__init__(f"{where}: {msg} [gs_status {status}]")


def load() -> C.CDLL:
    """Load libgpusort.so and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make` (hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, res, args in _PROTOS:
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if "GPUSORT_LIB" in os.environ:  # A/B runs against an older build: later entry points are absent
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def load_tuning() -> C.CDLL:
    """libgpusort_tuning.so (-DGS_TUNING: calibration kernels and tuning tile shapes; tools/ and bench.py's box_floor block —
    never the product path)."""
    path = os.path.join(_HERE, "lib", "libgpusort_tuning.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    lib.gs_debug_copy_floor.restype = _int
    lib.gs_debug_copy_floor.argtypes = [_vp, _vp, _u32, _u32, _u32, _vp]
    return lib


def check(status: int, where: str) -> None:
    if status != GS_OK:
        raise GpuSortError(status, where)
