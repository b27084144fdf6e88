"""The multi-GPU pipeline with the REAL per-rank engine (libgpusort.so kernels) on world_size 2 and 3 — on a
one-GPU box: the ranks share cuda:0 and talk over gloo (ShardedOneSweep stages the two collectives through host
memory for that backend; RCCL refuses two ranks on one device).  Everything else is the product path: top-byte
histogram + scan of the shard, splitters, stable split by destination, exchange, local 4-pass sort."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, shard, andc, pairs, q, bin_major=None, empty_rank=-1):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if bin_major is not None:
        # the local sorter is offered the two-level plan from 2^20 keys (default: 50 M keys / 2^25 + 1 pairs), so that buckets of test size
        # are landed top byte by top byte and sorted from the plan's second pass on (gs_mgpu_options::sorter through the harness's env)
        os.environ.update(GPUSORT_PLAN="2", GPUSORT_POS_MIN_LOG2="20", GPUSORT_MID_PATH="0")   # (mid path off: below 2^23 keys the two-launch route would take the bucket)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import gpusorting_amd as g
    from gpusorting_amd.sharded import ShardedOneSweep
    torch.cuda.set_device(0)
    keys = torch.empty(shard, dtype=torch.int32, device="cuda")
    slack = 3.0
    if andc < 0:  # one top byte holds ~90 % of the keys: forces the 12-bit prefix split
        g.init_random(keys, 10 + 1000 * rank, 0)
        pick = torch.empty(shard, dtype=torch.int32, device="cuda")
        g.init_random(pick, 99 + rank, 0)
        heavy = (pick.to(torch.int64) & 0xFFFFFFFF) % 10 != 0
        keys = torch.where(heavy, (keys & 0x00FFFFFF) | 0x5A000000, keys).contiguous()
        slack = 1.25 if world <= 4 else 1.5   # (world 8: whole 1/16ths of the heavy byte, 5.6 % of all keys each, against a share of 12.5 %)
    else:
        g.init_random(keys, 10 + 1000 * rank, andc)
    vals = (torch.arange(shard, dtype=torch.int32, device="cuda") + rank * shard) if pairs else None
    if rank == empty_rank:   # this rank brings nothing: it still takes part in every collective and receives its bucket
        keys, vals = keys[:0], (None if vals is None else vals[:0])
    k0 = keys.cpu().numpy().view(np.uint32).copy()
    v0 = None if vals is None else vals.cpu().numpy().view(np.uint32).copy()
    s = ShardedOneSweep(shard, slack=slack, pairs=pairs, value_bytes=4)   # the C++ pipeline (gs_onesweep_sort_sharded) over a host-staged transport
    bk, bv, nb = s.sort(keys, values=vals)
    torch.cuda.synchronize()
    s.check()   # gs_mgpu_check: no rank carried an error through the exchange, the local sorter is clean
    assert s.last_split == ("12-bit prefix" if andc < 0 else "top byte"), s.last_split
    if bin_major is not None:
        assert s.last_bin_major == (bin_major and nb > (1 << 20)), (s.last_bin_major, nb)
        tl = s.engine.sorter.last_plan()["two_level"] if nb > (1 << 20) else None
        assert tl is None or tl == (andc == 0), (tl, andc)   # uniform keys: the device runs the plan; preset 4: void -> copy + four LSD passes
    else:
        assert not s.last_bin_major
    q.put((rank, k0, v0, bk.cpu().numpy().view(np.uint32).copy(), None if bv is None else bv.cpu().numpy().view(np.uint32).copy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# world 8 (BASELINE.json configs[3]'s world size) on ONE GPU: eight ranks time-share cuda:0 over the host-staged transport
@pytest.mark.parametrize("world,andc,pairs,shard", [(2, 0, False, (1 << 20) + 77), (3, 0, True, 300001), (2, 1, True, (1 << 18) + 5),
                                                    (2, -1, False, (1 << 19) + 9), (3, -1, True, 200003),
                                                    (8, 0, False, (1 << 18) + 11), (8, 0, True, (1 << 18) + 3), (8, -1, True, 150001)])
def test_sharded_sort_real_engine_over_gloo(gpu, world, andc, pairs, shard):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shard, andc, pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=480) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    all_keys = np.concatenate([x[1] for x in got])
    out_keys = np.concatenate([x[3] for x in got])
    if not pairs:
        np.testing.assert_array_equal(out_keys, np.sort(all_keys))
    else:
        all_vals = np.concatenate([x[2] for x in got])
        out_vals = np.concatenate([x[4] for x in got])
        perm = np.argsort(all_keys, kind="stable")   # global stable order: (rank, position)
        np.testing.assert_array_equal(out_keys, all_keys[perm])
        np.testing.assert_array_equal(out_vals, all_vals[perm])
    for a, b in zip(got[:-1], got[1:]):              # buckets are contiguous ranges of the global order
        if a[3].size and b[3].size:
            assert a[3].max() <= b[3].min()


@pytest.mark.parametrize("world,andc,pairs,shard,empty_rank", [(2, 0, False, (3 << 19) + 77, -1), (3, 0, True, (3 << 19) + 1, -1), (2, 3, False, (3 << 20) + 5, -1),
                                                               (2, 3, True, (3 << 20) + 9, -1), (8, 0, False, (9 << 17) + 11, -1), (8, 0, True, (9 << 17) + 3, -1),
                                                               (3, 0, True, (7 << 18) + 7, 1), (3, 0, False, (7 << 18) + 7, 0)])
def test_sharded_sort_bucket_landed_bin_major(gpu, world, andc, pairs, shard, empty_rank):
    """Round 6: the bucket exchange goes one message per (peer, top byte) and a bucket that is offered the two-level
    plan is landed top byte by top byte in the local sort's alternate buffer — the sender's split was the top-byte partition — so the
    local sort starts at pass B.  World 2 / 3 / 8 with the real engine on one GPU; uniform keys (the plan runs) and entropy preset 4 (the
    device voids the plan: hy_void_copy_kernel + four LSD passes); keys-only and pairs with value = global index (stable across sources)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shard, andc, pairs, q, True, empty_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=480) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    all_keys = np.concatenate([x[1] for x in got])
    out_keys = np.concatenate([x[3] for x in got])
    if not pairs:
        np.testing.assert_array_equal(out_keys, np.sort(all_keys))
    else:
        all_vals = np.concatenate([x[2] for x in got])
        out_vals = np.concatenate([x[4] for x in got])
        perm = np.argsort(all_keys, kind="stable")   # global stable order: (rank, position)
        np.testing.assert_array_equal(out_keys, all_keys[perm])
        np.testing.assert_array_equal(out_vals, all_vals[perm])


def _failing_worker(rank, world, port, shard, where, pairs, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import gpusorting_amd as g
    from gpusorting_amd import _lib
    from gpusorting_amd.sharded import ShardedOneSweep
    torch.cuda.set_device(0)
    keys = torch.empty(shard, dtype=torch.int32, device="cuda")
    g.init_random(keys, 10 + 1000 * rank, 0)
    vals = torch.arange(shard, dtype=torch.int32, device="cuda") if pairs else None
    s = ShardedOneSweep(shard, slack=3.0, pairs=pairs, value_bytes=4)
    if rank == world - 1:
        s.debug_fail(where)            # this rank fails on its own in the next call
    outcome = []
    try:
        s.sort(keys, values=vals)
        outcome.append("sort-ok")
    except _lib.GpuSortError as e:
        outcome.append(f"sort-status-{e.status}")
    try:
        s.check()
        outcome.append("check-ok")
    except _lib.GpuSortError as e:
        outcome.append(f"check-status-{e.status}")
    # the context is still usable after a failure that everybody saw: a second, healthy call sorts
    bk, _, nb = s.sort(keys, values=vals)
    s.check()
    ok = bool((bk[1:nb].to(torch.int64) & 0xFFFFFFFF >= bk[:nb - 1].to(torch.int64) & 0xFFFFFFFF).all().item()) if nb > 1 else True
    total = torch.tensor([nb], dtype=torch.int64)
    dist.all_reduce(total)
    q.put((rank, outcome, ok and int(total.item()) == shard * world))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("where,pairs", [(1, False), (2, False), (2, True)])
def test_a_rank_that_fails_alone_does_not_strand_its_peers(gpu, where, pairs):
    """Round-2 review, multi-GPU item (c): a rank that fails alone — before the histogram gather (where = 1) or after the plan
    (where = 2: the peers are already committed to the exchange) — must not leave the others inside a collective.  Two ranks on
    one GPU over the host-staged transport; the last rank fails.  where = 1: its row of the gather is poisoned, EVERY rank's
    sort returns at once (the failing one with its own error, the other with GS_ERR_COMM).  where = 2: the failing rank still
    serves the exchange and reports in the closing status gather: its own sort returns GS_ERR_HIP, the peer's sort returns
    GS_OK and its gs_mgpu_check says GS_ERR_COMM.  Nobody hangs, and the next call on the same contexts sorts."""
    from gpusorting_amd import _lib
    world, shard = 2, (1 << 18) + 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, shard, where, pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    healthy, failing = got[0][1], got[1][1]
    if where == 1:
        assert healthy[0] == f"sort-status-{_lib.GS_ERR_COMM}", got
        assert failing[0] == f"sort-status-{_lib.GS_ERR_HIP}", got
    else:
        assert healthy == ["sort-ok", f"check-status-{_lib.GS_ERR_COMM}"], got
        assert failing[0] == f"sort-status-{_lib.GS_ERR_HIP}", got
    assert got[0][2] and got[1][2], got   # the follow-up call on the same contexts is exact in size and sorted


# ---- the REAL transport: RCCL over xGMI, one GPU per rank.  Lights up by itself on a box with two GPUs or more -----------------
def _rccl_worker(rank, world, port, shard, pairs, alltoallv, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    sys.path.insert(0, ROOT)
    import gpusorting_amd as g
    from gpusorting_amd.sharded import ShardedOneSweep
    keys = torch.empty(shard, dtype=torch.int32, device="cuda")
    g.init_random(keys, 10 + 1000 * rank, 0)
    vals = (torch.arange(shard, dtype=torch.int32, device="cuda") + rank * shard) if pairs else None
    if rank == empty_rank:   # this rank brings nothing: it still takes part in every collective and receives its bucket
        keys, vals = keys[:0], (None if vals is None else vals[:0])
    k0 = keys.cpu().numpy().view(np.uint32).copy()
    v0 = None if vals is None else vals.cpu().numpy().view(np.uint32).copy()
    s = ShardedOneSweep(shard, pairs=pairs, value_bytes=4)   # gs_mgpu_create_ex: ncclCommInitRank (+ ncclCommSplit for the values)
    s.set_alltoallv(alltoallv)
    outs = []
    for rep in range(2):                                     # the context's second call as well
        bk, bv, nb = s.sort(keys, values=vals)
        torch.cuda.synchronize()
        s.check()
        outs.append((bk[:nb].cpu().numpy().view(np.uint32).copy(), None if bv is None else bv[:nb].cpu().numpy().view(np.uint32).copy()))
    q.put((rank, k0, v0, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("alltoallv", [False, True])
def test_sharded_sort_over_rccl_one_gpu_per_rank(gpu, pairs, alltoallv):
    """the RCCL exchange (grouped ncclSend / ncclRecv and ncclAllToAllv), the values' second communicator and
    stream, and the closing status gather with a real peer.  Needs two GPUs: skipped on the one-GPU boxes of this pool."""
    import torch
    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("one GPU visible: RCCL refuses two ranks on one device (the gloo-staged tests above cover the pipeline)")
    shard = (1 << 22) + 12345
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, shard, pairs, alltoallv, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    all_keys = np.concatenate([x[1] for x in got])
    perm = np.argsort(all_keys, kind="stable")
    for rep in range(2):
        out_keys = np.concatenate([x[3][rep][0] for x in got])
        np.testing.assert_array_equal(out_keys, all_keys[perm])
        if pairs:
            all_vals = np.concatenate([x[2] for x in got])
            np.testing.assert_array_equal(np.concatenate([x[3][rep][1] for x in got]), all_vals[perm])
