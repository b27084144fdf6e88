"""Multi-GPU sharded sort: one MSD split + RCCL all-to-all-v + per-GPU OneSweep.

No reference counterpart (the reference is single-GPU); this is
BASELINE.json configs[3].  One process per GPU, ``torch.distributed`` (backend
"nccl" == RCCL over xGMI).  Steps on every rank, all on the rank's own data:

  1. top-byte histogram of the local shard (the GlobalHistogram kernel's row 3)
  2. ONE small all_gather of the 256-bin histograms -> every rank knows every
     (source, bin) count, hence the global histogram, the splitters
     (``gs_msd_splitters_n``: equal-count buckets at top-byte granularity) and all
     send/receive counts without a second exchange.  If a bucket would not fit
     its rank (skewed keys), the split is redone at 12-bit prefix granularity:
     4096-bin histograms, shard ordered by its top two bytes
  3. a stable DigitBinningPass on the top byte groups the shard by destination
  4. ``all_to_all_single`` with split sizes (RCCL AllToAllv; point-to-point on
     all xGMI links at once) moves every key to its owner
  5. the local 4-pass OneSweep sorts the received bucket
Result: rank r holds the r-th contiguous range of the globally sorted array
(concatenation over ranks == the sorted whole).  Received data is ordered by
(source rank, source position), so for pairs the whole pipeline is stable.

The local engine is injected (``HipLocalEngine`` is the product; the CPU tests
inject an oracle-backed engine and run the same control flow over gloo).
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


class HipLocalEngine:
    """Per-rank work on the GPU through libgpusort.so."""

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


class ShardedOneSweep:
    def __init__(self, shard_keys: int, engine=None, group=None, slack: float = 1.25, pairs: bool = False,
                 value_bytes: int = 4, always_exchange: bool = False):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.shard_keys = int(shard_keys)
        self.capacity = min(int(shard_keys * slack) + 256, _lib.GS_MAX_KEYS)
        self.pairs = pairs
        self.always_exchange = always_exchange  # run the split/exchange path even for one rank (single-GPU tests)
        self.engine = engine if engine is not None else HipLocalEngine(self.capacity, pairs, value_bytes)
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
        self.last_counts = None
        self.last_split = None    # "top byte" or "12-bit prefix"
        # gloo cannot move device memory: with that backend and a GPU engine the two collectives are staged
        # through host tensors (test configuration: several ranks sharing one GPU; the product backend is RCCL)
        self._host_staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"

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
        eng, W = self.engine, self.world
        if W == 1 and not self.always_exchange:
            self._recv[:n].copy_(keys[:n])
            if values is not None:
                self._recv_v[:n].copy_(values[:n])
            eng.sort(self._recv, n, self._recv_v if values is not None else None)
            return self._recv[:n], (self._recv_v[:n] if values is not None else None), n

        # 1-2: histograms of every rank, splitters, split sizes
        local = torch.from_numpy(eng.top_byte_histogram(keys, n)).to(self._gather.device)
        self._all_gather(self._gather, local)
        table = self._gather.cpu().numpy().reshape(W, 256)          # [source, top byte]

        def split(tab):
            first_bin = msd_splitters(tab.sum(axis=0).astype(np.uint64), W)
            csum = np.concatenate([np.zeros((W, 1), np.int64), np.cumsum(tab, axis=1)], axis=1)
            return csum[:, first_bin[1:]] - csum[:, first_bin[:-1]]  # [source, dest]

        per_dest = split(table)
        fine = int(per_dest.sum(axis=0).max()) > self.capacity       # same table on every rank: same decision
        if fine:
            # Some rank's top-byte bucket would not fit (skewed keys: SURVEY.md §8e).  Split at the 12-bit prefix
            # instead: 4096-bin histograms, and the shard ordered by its top TWO bytes so that every prefix
            # range is contiguous (one more partition pass; only on this path).
            if self._gather_fine is None:
                self._gather_fine = torch.empty(W * 4096, dtype=torch.int64, device=self._gather.device)
            local = torch.from_numpy(eng.fine_histogram(keys, n)).to(self._gather.device)
            self._all_gather(self._gather_fine, local)
            per_dest = split(self._gather_fine.cpu().numpy().reshape(W, 4096))
        self.last_split = "12-bit prefix" if fine else "top byte"
        send = per_dest[self.rank].tolist()
        recv = per_dest[:, self.rank].tolist()
        n_recv = int(sum(recv))
        self.last_counts = (send, recv)
        if int(per_dest.sum(axis=0).max()) > self.capacity:          # every rank raises together
            raise RuntimeError(f"rank {self.rank}: a bucket of {int(per_dest.sum(axis=0).max())} keys exceeds capacity "
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
