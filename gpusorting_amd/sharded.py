"""Multi-GPU sharded sort: one MSD split + RCCL bucket exchange + per-GPU OneSweep.

No reference counterpart (the reference is single-GPU); this is
BASELINE.json configs[3].  One process per GPU.  The PRODUCT pipeline lives behind
the C-ABI (``gs_onesweep_sort_sharded`` in libgpusort.so: C++ calling RCCL directly —
histogram, all-gather, splitters and counts on the device, one partition pass,
keys and values exchanged in ONE group of send/recv pairs, local 4-pass sort; one
host wait on a few dozen counts).  ``ShardedOneSweep`` here is its binding:

  * backend "nccl" (RCCL over xGMI): rank 0 draws a unique id
    (``gs_mgpu_get_unique_id``), ``torch.distributed`` only carries those 128
    bytes to the other ranks, and every rank creates its ``gs_mgpu`` context;
  * backend "gloo" with a GPU (several ranks sharing ONE GPU: the test
    configuration): the same C++ pipeline runs over a host-staged transport
    injected through ``gs_mgpu_create_with_transport``.

Rank r ends up with the r-th contiguous range of the globally sorted array
(concatenation over ranks == the sorted whole); received data is ordered by
(source rank, source position), so for pairs the whole pipeline is stable.

With an injected ``engine`` (the CPU tests inject an oracle-backed one and run
over gloo without any GPU) the same steps are driven from Python: the exchange
plan still comes from the C-ABI (``gs_msd_plan``, the host twin of the device
kernel), the collectives from ``torch.distributed``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def msd_splitters(hist: np.ndarray, world: int) -> np.ndarray:
    """first_bin[r] = first bin (top byte, or 12-bit prefix for a 4096-bin histogram) rank r owns; host-side C-ABI call."""
    h = np.ascontiguousarray(hist, dtype=np.uint64)
    fb = (C.c_uint32 * (world + 1))()
    _lib.check(_lib.load().gs_msd_splitters_n(h.ctypes.data_as(C.POINTER(C.c_uint64)), h.size, world, fb), "gs_msd_splitters_n")
    return np.frombuffer(fb, dtype=np.uint32).copy()


def msd_plan(table: np.ndarray, rank: int, capacity: int) -> dict:
    """The exchange plan of one rank from the gathered [source, bin] table (gs_msd_plan: the host twin of the device
    kernel): splitters, what this rank sends to / receives from every peer, the largest bucket and whether it fits."""
    t = np.ascontiguousarray(table, dtype=np.uint32)
    world, nbins = t.shape
    plan = np.zeros(4 + 3 * world + 1, dtype=np.uint32)
    _lib.check(_lib.load().gs_msd_plan(t.ctypes.data_as(C.POINTER(C.c_uint32)), nbins, world, rank, min(int(capacity), 0xFFFFFFFF),
                                       plan.ctypes.data_as(C.POINTER(C.c_uint32))), "gs_msd_plan")
    return _plan_dict(plan, world)


def _plan_dict(plan: np.ndarray, world: int) -> dict:
    return {"n_recv": int(plan[0]), "overflow": bool(plan[1]), "max_bucket": int(plan[2]),
            "send": plan[4:4 + world].astype(np.int64).tolist(), "recv": plan[4 + world:4 + 2 * world].astype(np.int64).tolist(),
            "first_bin": plan[4 + 2 * world:4 + 3 * world + 1].copy()}


class HipLocalEngine:
    """Per-rank work on the GPU through libgpusort.so (the Python-driven pipeline; the product uses gs_mgpu)."""

    def __init__(self, capacity: int, pairs: bool = False, value_bytes: int = 4, key_type: int = 0):
        from .onesweep import MODE_KEYS_ONLY, MODE_PAIRS, OneSweep
        self.sorter = OneSweep(capacity, key_type=key_type, mode=MODE_PAIRS if pairs else MODE_KEYS_ONLY,
                               value_bytes=value_bytes if pairs else 0)
        self.device = self.sorter.device

    def empty_like_keys(self, n):
        return torch.empty(n, dtype=torch.int32, device=self.device)

    def top_byte_histogram(self, keys, n) -> np.ndarray:
        # one histogram + scan of the shard serves both the split decision and the partition pass
        self._prepared = (keys.data_ptr(), n)
        return self.sorter.msd_prepare(keys, n).astype(np.int64)

    def partition_by_top_byte(self, keys, out, n, values=None, values_out=None):
        if getattr(self, "_prepared", None) == (keys.data_ptr(), n):
            self._prepared = None
            self.sorter.msd_partition(keys, out, n=n, values_in=values, values_out=values_out)
        else:
            self.sorter.digit_pass(keys, out, 3, n=n, values_in=values, values_out=values_out)

    def fine_histogram(self, keys, n) -> np.ndarray:
        return self.sorter.msd_fine_histogram(keys, n).astype(np.int64)

    def partition_by_top12(self, keys, out, tmp, n, values=None, values_out=None, values_tmp=None):
        """Stable order by the top two bytes (pass 2 into tmp, pass 3 into out): contiguous in the 12-bit prefix."""
        self._prepared = None
        self.sorter.digit_pass(keys, tmp, 2, n=n, values_in=values, values_out=values_tmp)
        self.sorter.digit_pass(tmp, out, 3, n=n, values_in=values_tmp, values_out=values_out)

    def sort(self, keys, n, values=None):
        self.sorter.sort(keys, values, n=n)

    def synchronize(self):
        torch.cuda.synchronize()


class _NativeEngine:
    """What bench.py and the tests look at when the C++ pipeline runs: the local sorter inside the gs_mgpu context."""

    def __init__(self, sorter, device):
        self.sorter = sorter
        self.device = device

    def synchronize(self):
        torch.cuda.synchronize()


class _HostStagedTransport:
    """gs_mgpu_transport over a gloo group for ranks that share one GPU (tests): device buffers are staged through
    host memory with the HIP runtime, the collectives are torch.distributed point-to-point / all_gather calls."""

    def __init__(self, group, rank, world):
        self.group, self.rank, self.world = group, rank, world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self._ag = _lib.ALL_GATHER_FN(self._all_gather)
        self._ex = _lib.EXCHANGE_FN(self._exchange)
        self.struct = _lib.MgpuTransport(None, self._ag, self._ex)

    def _d2h(self, ptr, nbytes, stream):
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8)
        if nbytes:
            self.hip.hipStreamSynchronize(stream)
            if self.hip.hipMemcpy(buf.data_ptr(), ptr, nbytes, 2) != 0:
                raise RuntimeError("hipMemcpy D2H failed")
        return buf[:nbytes]

    def _h2d(self, ptr, t):
        if t.numel() and self.hip.hipMemcpy(ptr, t.data_ptr(), t.numel(), 1) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    def _all_gather(self, user, d_send, d_recv, count, stream):
        try:
            mine = self._d2h(d_send, count * 4, stream)
            parts = [torch.empty(count * 4, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, mine.contiguous(), group=self.group)
            self._h2d(d_recv, torch.cat(parts))
            return 0
        except Exception:  # noqa: BLE001 — an exception must not cross the C boundary
            return 1

    def _exchange(self, user, n_arrays, d_send, d_recv, elem_bytes, send_counts, send_displs, recv_counts, recv_displs, stream):
        try:
            W, me = self.world, self.rank
            for a in range(n_arrays):
                eb = elem_bytes[a]
                outs = [torch.empty(recv_counts[p] * eb, dtype=torch.uint8) for p in range(W)]
                reqs = []
                for p in range(W):
                    # (piece by piece: the library also calls once per top byte — one message per (peer, top byte), gpusort_mgpu.hpp —
                    #  and then a peer's piece lies anywhere in the send buffer)
                    piece = self._d2h((d_send[a] or 0) + send_displs[p] * eb, send_counts[p] * eb, stream)
                    if p == me:
                        outs[p].copy_(piece)
                    else:
                        reqs.append(dist.isend(piece.contiguous(), p, group=self.group))
                        reqs.append(dist.irecv(outs[p], p, group=self.group))
                for r in reqs:
                    r.wait()
                for p in range(W):
                    self._h2d(d_recv[a] + recv_displs[p] * eb, outs[p])
            return 0
        except Exception:  # noqa: BLE001
            return 1


class ShardedOneSweep:
    def __init__(self, shard_keys: int, engine=None, group=None, slack: float = 1.25, pairs: bool = False,
                 value_bytes: int = 4, always_exchange: bool = False, key_type: int = 0):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.shard_keys = int(shard_keys)
        self.capacity = min(int(shard_keys * slack) + 256, _lib.GS_MAX_KEYS)
        self.pairs = pairs
        self.value_bytes = value_bytes if pairs else 0
        self.key_type = key_type
        self.always_exchange = always_exchange  # run the split/exchange path even for one rank (single-GPU tests)
        self.last_counts = None
        self.last_split = None    # "top byte" or "12-bit prefix"
        self.last_bin_major = False  # gs_mgpu_last_layout of the last native sort
        self._ctx = None
        self._transport = None
        if engine is None:
            self._native_init()
            return
        self.engine = engine
        dev = self.engine.device
        self._part = self.engine.empty_like_keys(self.shard_keys)
        self._recv = self.engine.empty_like_keys(self.capacity)
        self._part_v = self._recv_v = None
        if pairs:
            dt = torch.int32 if value_bytes == 4 else torch.int64
            self._part_v = torch.empty(self.shard_keys, dtype=dt, device=dev)
            self._recv_v = torch.empty(self.capacity, dtype=dt, device=dev)
        self._gather = torch.empty(self.world * 256, dtype=torch.int64, device=dev)
        self._gather_fine = None  # world x 4096, allocated on first use (skewed shards only)
        # gloo cannot move device memory: with that backend and a GPU engine the two collectives are staged
        # through host tensors
        self._host_staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"

    # ---- the product: gs_mgpu behind the C-ABI -------------------------------------------------------------
    def _native_init(self):
        from .onesweep import MODE_KEYS_ONLY, MODE_PAIRS, OneSweep
        if not torch.cuda.is_available():
            raise RuntimeError("gpusorting_amd needs a GPU: the product path has no CPU fallback")
        lib = _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())
        mode = MODE_PAIRS if self.pairs else MODE_KEYS_ONLY
        ctx = C.c_void_p()
        if dist.get_backend(self.group) == "gloo":
            self._transport = _HostStagedTransport(self.group, self.rank, self.world)
            mopts = _lib.mgpu_options_from_env()
            st = lib.gs_mgpu_create_with_transport_ex(C.byref(ctx), C.byref(self._transport.struct), self.rank, self.world,
                                                      self.shard_keys, self.capacity, mode, self.value_bytes, C.byref(mopts))
            _lib.check(st, "gs_mgpu_create_with_transport_ex")
        else:
            uid = torch.zeros(_lib.GS_MGPU_UNIQUE_ID_BYTES, dtype=torch.uint8)
            if self.rank == 0:
                buf = (C.c_uint8 * _lib.GS_MGPU_UNIQUE_ID_BYTES)()
                _lib.check(lib.gs_mgpu_get_unique_id(buf), "gs_mgpu_get_unique_id")
                uid = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
            uid = uid.to(dev)
            src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
            dist.broadcast(uid, src, group=self.group)  # the only thing torch.distributed carries: 128 bytes, once
            raw = (C.c_uint8 * _lib.GS_MGPU_UNIQUE_ID_BYTES)(*uid.cpu().tolist())
            mopts = _lib.mgpu_options_from_env()
            _lib.check(lib.gs_mgpu_create_ex(C.byref(ctx), raw, self.rank, self.world, self.shard_keys, self.capacity, mode,
                                             self.value_bytes, C.byref(mopts)), "gs_mgpu_create_ex")
        self._ctx = ctx
        if self.always_exchange:
            _lib.check(lib.gs_mgpu_set_force_exchange(ctx, 1), "gs_mgpu_set_force_exchange")
        sorter = OneSweep._borrow(lib.gs_mgpu_sorter(ctx), self.capacity, mode, self.value_bytes, self.key_type)
        self.engine = _NativeEngine(sorter, dev)
        self._recv = torch.empty(self.capacity, dtype=torch.int32, device=dev)
        self._recv_v = None
        if self.pairs:
            self._recv_v = torch.empty(self.capacity, dtype=torch.int32 if self.value_bytes == 4 else torch.int64, device=dev)

    def close(self):
        if self._ctx:
            _lib.load().gs_mgpu_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def check(self) -> None:
        """gs_mgpu_check: synchronises and raises if the last call failed on ANY rank (a peer that failed on its own after the
        plan has still served this rank's receives, and says so in the call's closing all-gather) or on the local sorter."""
        from .onesweep import _stream_ptr
        if self._ctx:
            _lib.check(_lib.load().gs_mgpu_check(self._ctx, _stream_ptr()), "gs_mgpu_check")

    def debug_fail(self, where: int) -> None:
        """Test hook: this rank's next sort fails on its own (1: before the histogram gather, 2: after the plan)."""
        _lib.check(_lib.load().gs_mgpu_debug_fail(self._ctx, int(where)), "gs_mgpu_debug_fail")

    def set_alltoallv(self, on: bool) -> None:
        """The RCCL transport's bucket exchange: ncclAllToAllv (True) or grouped ncclSend / ncclRecv (False, default).  Every rank
        must choose alike; takes effect from the next sort."""
        if self._ctx:
            _lib.check(_lib.load().gs_mgpu_set_alltoallv(self._ctx, 1 if on else 0), "gs_mgpu_set_alltoallv")

    def profile(self) -> dict:
        """Phase times (ms) and off-rank bytes of the last native sort on this rank."""
        ms = (C.c_float * 4)()
        sent, recv, fine = C.c_uint64(), C.c_uint64(), C.c_uint32()
        _lib.check(_lib.load().gs_mgpu_get_profile(self._ctx, ms, C.byref(sent), C.byref(recv), C.byref(fine)), "gs_mgpu_get_profile")
        return {"split_ms": ms[0], "exchange_ms": ms[1], "local_sort_ms": ms[2], "total_ms": ms[3],
                "bytes_sent": int(sent.value), "bytes_received": int(recv.value), "fine_split": bool(fine.value)}

    def _native_sort(self, keys, n, values):
        from .onesweep import _stream_ptr
        lib = _lib.load()
        out_n = C.c_uint32()
        st = lib.gs_onesweep_sort_sharded(self._ctx, keys.data_ptr(), None if values is None else values.data_ptr(), n,
                                          self.key_type, self._recv.data_ptr(),
                                          None if values is None else self._recv_v.data_ptr(), C.byref(out_n), _stream_ptr())
        if st == _lib.GS_ERR_SIZE:
            raise RuntimeError(f"rank {self.rank}: a bucket exceeds capacity {self.capacity} even at 12-bit prefix granularity "
                               f"(raise slack)")
        _lib.check(st, "gs_onesweep_sort_sharded")
        nr = int(out_n.value)
        if self.world > 1 or self.always_exchange:
            plan = np.zeros(4 + 3 * self.world + 1, dtype=np.uint32)
            _lib.check(lib.gs_mgpu_last_plan(self._ctx, plan.ctypes.data_as(C.POINTER(C.c_uint32)), plan.size), "gs_mgpu_last_plan")
            d = _plan_dict(plan, self.world)
            self.last_counts = (d["send"], d["recv"])
            self.last_split = "12-bit prefix" if int(plan[3]) else "top byte"  # (profile() would wait for the local sort)
            bm = C.c_uint32(0)
            _lib.check(lib.gs_mgpu_last_layout(self._ctx, C.byref(bm)), "gs_mgpu_last_layout")
            self.last_bin_major = bool(bm.value)  # the bucket was landed top byte by top byte: the local sort skipped its top-byte pass
        return self._recv[:nr], (self._recv_v[:nr] if values is not None else None), nr

    # ---- the same steps driven from Python over an injected engine (CPU tests) ------------------------------------
    def _all_gather(self, out, inp):
        if not self._host_staged:
            dist.all_gather_into_tensor(out, inp, group=self.group)
            return
        parts = [torch.empty(inp.numel(), dtype=inp.dtype) for _ in range(self.world)]
        dist.all_gather(parts, inp.cpu(), group=self.group)
        out.copy_(torch.cat(parts))

    def _all_to_all(self, out, inp, recv, send):
        if not self._host_staged:
            dist.all_to_all_single(out, inp, recv, send, group=self.group)
            return
        src = inp.cpu()
        outs = [torch.empty(r, dtype=inp.dtype) for r in recv]
        ins = list(torch.split(src, send))
        # gloo has no all_to_all: point-to-point, lower rank sends first
        reqs = []
        for peer in range(self.world):
            if peer == self.rank:
                outs[peer].copy_(ins[peer])
            else:
                reqs.append(dist.isend(ins[peer].contiguous(), peer, group=self.group))
                reqs.append(dist.irecv(outs[peer], peer, group=self.group))
        for r in reqs:
            r.wait()
        if out.numel():
            out.copy_(torch.cat(outs))

    def sort(self, keys: torch.Tensor, n: int | None = None, values: torch.Tensor | None = None):
        """Sort the distributed array whose local shard is ``keys[:n]``.

        Returns ``(bucket_keys, bucket_values_or_None, n_bucket)``: views into this
        object's receive buffers holding this rank's range of the global result.
        """
        n = keys.numel() if n is None else int(n)
        if n > keys.numel() or n > self.shard_keys or (values is not None and n > values.numel()):
            raise ValueError(f"n = {n} exceeds the shard buffers (keys {keys.numel()}, shard capacity {self.shard_keys})")
        if (values is not None) != self.pairs:
            raise ValueError("values must be given exactly when the object was built with pairs=True")
        if self._ctx:
            return self._native_sort(keys, n, values)
        eng, W = self.engine, self.world
        if W == 1 and not self.always_exchange:
            self._recv[:n].copy_(keys[:n])
            if values is not None:
                self._recv_v[:n].copy_(values[:n])
            eng.sort(self._recv, n, self._recv_v if values is not None else None)
            return self._recv[:n], (self._recv_v[:n] if values is not None else None), n

        # 1-2: histograms of every rank -> the plan (splitters, split sizes, overflow): gs_msd_plan
        local = torch.from_numpy(eng.top_byte_histogram(keys, n)).to(self._gather.device)
        self._all_gather(self._gather, local)
        plan = msd_plan(self._gather.cpu().numpy().reshape(W, 256), self.rank, self.capacity)  # [source, top byte]
        fine = plan["overflow"]                                      # same table on every rank: same decision
        if fine:
            # Some rank's top-byte bucket would not fit (skewed keys: SURVEY.md §8e).  Split at the 12-bit prefix
            # instead: 4096-bin histograms, and the shard ordered by its top TWO bytes so that every prefix
            # range is contiguous (one more partition pass; only on this path).
            if self._gather_fine is None:
                self._gather_fine = torch.empty(W * 4096, dtype=torch.int64, device=self._gather.device)
            local = torch.from_numpy(eng.fine_histogram(keys, n)).to(self._gather.device)
            self._all_gather(self._gather_fine, local)
            plan = msd_plan(self._gather_fine.cpu().numpy().reshape(W, 4096), self.rank, self.capacity)
        self.last_split = "12-bit prefix" if fine else "top byte"
        send, recv, n_recv = plan["send"], plan["recv"], plan["n_recv"]
        self.last_counts = (send, recv)
        if plan["overflow"]:                                          # every rank raises together
            raise RuntimeError(f"rank {self.rank}: a bucket of {plan['max_bucket']} keys exceeds capacity "
                               f"{self.capacity} even at 12-bit prefix granularity (raise slack)")
        # 3: group by destination (stable)
        if fine:
            eng.partition_by_top12(keys, self._part, self._recv[:n], n, values, self._part_v,
                                   None if values is None else self._recv_v[:n])
        else:
            eng.partition_by_top_byte(keys, self._part, n, values, self._part_v)
        # 4: bucket exchange
        self._all_to_all(self._recv[:n_recv], self._part[:n], recv, send)
        if values is not None:
            self._all_to_all(self._recv_v[:n_recv], self._part_v[:n], recv, send)
        # 5: local sort
        if n_recv:
            eng.sort(self._recv, n_recv, self._recv_v if values is not None else None)
        return self._recv[:n_recv], (self._recv_v[:n_recv] if values is not None else None), n_recv
