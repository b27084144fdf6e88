"""GPU parity tests (-m gpu): the HIP path through the C-ABI vs the CPU oracle,
bit-exact (integer/byte/index work: no tolerance)."""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "onesweep_golden.npz")
_NP_KEY = {0: np.uint32, 1: np.int32, 2: np.float32}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def to_dev(a):
    import torch
    if a.dtype == np.uint64:
        return torch.from_numpy(a.view(np.int64)).cuda()
    return torch.from_numpy(a.view(np.int32)).cuda()


def to_host(t, dtype):
    return t.cpu().numpy().view(dtype)


@pytest.fixture(autouse=True)
def _every_test_under_three_routings(routing):
    """(tests/conftest.py: the library's own routing, the mid-size route off, forced position chains)"""
    return routing


@pytest.fixture(scope="module")
def P(gpu):
    s = gpu.OneSweep(1 << 16)
    p = s.partition_size
    s.close()
    return p


def _gpu_sort(gpu, keys, kt=0, order=0, vals=None, max_keys=None):
    mode = gpu.MODE_KEYS_ONLY if vals is None else gpu.MODE_PAIRS
    vb = 0 if vals is None else vals.dtype.itemsize
    s = gpu.OneSweep(max_keys or max(keys.size, 1), order, kt, mode, vb)
    dk = to_dev(keys)
    dv = None if vals is None else to_dev(vals)
    s.sort(dk, dv)
    s.check()
    _assert_scan_state(s, keys.size)   # every sort of every test leaves the chained-scan state as it must be
    out = to_host(dk, np.uint32), (None if vals is None else to_host(dv, vals.dtype))
    s.close()
    return out


def _assert_scan_state(s, n):
    """gs_debug_check_state after a completed sort: every descriptor row INCLUSIVE and non-decreasing along its
    chain, every chain's tickets >= its tiles, the histogram region handed back zeroed, and every pass that ran
    accounts for exactly n keys (a dropped identity pass for none) — cf. UtilityKernels.cuh:482-502."""
    r = s.check_state()
    if "fault" in os.path.basename(os.environ.get("GPUSORT_LIB", "")):
        # fault-injection builds (tools/r03_run25.sh: the fuzz sweep under libgpusort_fault.so): one tile per pass never publishes
        # its rows — its successors' recount leaves them REDUCTION (a tile count, below the inclusive count in front of it) —
        # so only the ticket and histogram invariants hold
        assert r["chains_short_of_tickets"] == 0 and r["hist_words_nonzero"] == 0, r
        return
    assert r["rows_not_inclusive"] == 0 and r["rows_not_monotone"] == 0, r
    assert r["chains_short_of_tickets"] == 0 and r["hist_words_nonzero"] == 0, r
    assert all(k in (0, n) for k in r["keys_per_pass"]), (n, r)


def test_init_random_parity(gpu, oracle):
    import torch
    for n, seed, andc, vb in [(1, 1, 0, 0), (63, 2, 0, 4), (65536, 10, 0, 0), (65537, 10, 1, 8), (200003, 77, 4, 4),
                              (1 << 20, 26, 0, 0)]:
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        dv = None if not vb else torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
        gpu.init_random(dk, seed, andc, dv)
        torch.cuda.synchronize()
        ref = oracle.init_random(n, seed, andc, vb)
        rk, rv = (ref, None) if not vb else ref
        np.testing.assert_array_equal(to_host(dk, np.uint32), rk)
        if vb:
            np.testing.assert_array_equal(to_host(dv, rv.dtype), rv)


def test_init_random_matches_reference_produced_vectors(gpu):
    """The HIP generator against keys written by the reference's OWN InitRandom kernels (run on a CPU from the
    reference sources: tests/golden/make_ref_golden.py -> tests/golden/ref_init_random.npz)."""
    import torch
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_init_random.npz"))
    for i, (n, seed, andc) in enumerate(g["cases"].tolist()):
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        dv = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.init_random(dk, seed, andc, dv)
        torch.cuda.synchronize()
        k = to_host(dk, np.uint32)
        assert crc(k) == int(g[f"crc_{i}"]), (n, seed, andc)
        np.testing.assert_array_equal(k[:64], g[f"head_{i}"])
        np.testing.assert_array_equal(k[-64:], g[f"tail_{i}"])
        np.testing.assert_array_equal(to_host(dv, np.uint32), k)


def test_gpu_matches_reference_kernel_vectors(gpu):
    """The HIP path against outputs of the reference's OWN OneSweep kernels (GPUSortingCUDA/Sort/OneSweep.cu run on
    a CPU by the SIMT emulator of oracle/shim; tests/golden/ref_onesweep.npz): the global histogram, the key (and
    payload) buffer after each of the four passes, and the sorted result of the complete sort."""
    import torch
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_onesweep.npz"))
    for i, (n, seed, andc, pairs) in enumerate(g["cases"].tolist()):
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.init_random(dk, seed, andc)
        dv = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
        s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if pairs else gpu.MODE_KEYS_ONLY, value_bytes=4 if pairs else 0)
        assert gpu.validate(dk) == int(g[f"verr_{i}"]), (n, seed)   # the reference's Validate on the unsorted input
        assert crc(s.global_histogram(dk).astype(np.uint32)) == int(g[f"hist_{i}"]), (n, seed)
        ck, cv = dk.clone(), (None if dv is None else dv.clone())
        for p in range(4):                                  # pass by pass
            ok = torch.zeros_like(ck)
            ov = None if cv is None else torch.zeros_like(cv)
            s.digit_pass(ck, ok, p, values_in=cv, values_out=ov)
            s.check()
            assert crc(to_host(ok, np.uint32)) == int(g[f"kcrc_{i}"][p]), (n, seed, p)
            if pairs:
                assert crc(to_host(ov, np.uint32)) == int(g[f"vcrc_{i}"][p]), (n, seed, p)
            ck, cv = ok, ov
        for small in (True, False):                         # the complete sort, both small-n routes
            s.set_small_path(small)
            s.set_mid_path(small)
            fk, fv = dk.clone(), (None if dv is None else dv.clone())
            s.sort(fk, fv)
            s.check()
            out = to_host(fk, np.uint32)
            assert crc(out) == int(g[f"kcrc_{i}"][3]), (n, seed, small)
            np.testing.assert_array_equal(out[:32], g[f"head_{i}"])
            np.testing.assert_array_equal(out[-32:], g[f"tail_{i}"])
            if pairs:
                assert crc(to_host(fv, np.uint32)) == int(g[f"vcrc_{i}"][3]), (n, seed, small)
        s.close()


@pytest.mark.parametrize("kt", [0, 1, 2])
def test_global_histogram_parity(gpu, oracle, kt):
    for n in (1, 3, 4, 5, 1023, 65536, 65539, (1 << 22) + 1):
        keys = oracle.init_random(n, n + 1, 0)
        s = gpu.OneSweep(n, key_type=kt)
        h = s.global_histogram(to_dev(keys))
        s.close()
        np.testing.assert_array_equal(h, oracle.global_histogram(keys, kt), err_msg=f"n={n}")


@pytest.mark.parametrize("kt", [0, 1, 2])
def test_scan_digit_starts_parity(gpu, oracle, kt):
    """SURVEY.md 8a row A2, directly: the digit starts the Scan kernel leaves in the first descriptor row of each pass, read back
    and diffed against the oracle's restatement of the reference's Scan (GPUSortingCUDA/Sort/OneSweep.cu:125-162; the store of
    :141-158 is (exclusive prefix << 2) | FLAG_INCLUSIVE).  Uniform and skewed keys, sizes on both sides of the tile borders."""
    for n, andc in ((1, 0), (5, 0), (1023, 0), (65536, 0), (65539, 2), ((1 << 22) + 1, 0), ((1 << 24) + 3, 4)):
        keys = oracle.init_random(n, n + 7, andc)
        s = gpu.OneSweep(n, key_type=kt)
        rows = s.scan_rows(to_dev(keys))
        s.close()
        assert ((rows & 3) == 2).all(), f"n={n}: a seed row is not INCLUSIVE"
        np.testing.assert_array_equal(rows >> 2, oracle.scan(oracle.global_histogram(keys, kt)).reshape(4, 256), err_msg=f"n={n}")


@pytest.mark.parametrize("vb", [0, 4, 8])
def test_each_digit_pass_parity(gpu, oracle, P, vb):
    import torch
    for n in (100, P, P + 1, 3 * P + 17, (1 << 20) + 5):
        keys = oracle.init_random(n, n, 1)
        vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, value_bytes=vb)
        dk = to_dev(keys)
        dv = None if not vb else to_dev(vals)
        for p in range(4):
            for rev in (False, True):
                ok = torch.zeros_like(dk)
                ov = None if not vb else torch.zeros_like(dv)
                s.digit_pass(dk, ok, p, values_in=dv, values_out=ov, reverse_index=rev)
                s.check()
                ref = oracle.digit_pass(keys, 8 * p, 0, vals, rev)
                rk, rv = (ref, None) if not vb else ref
                np.testing.assert_array_equal(to_host(ok, np.uint32), rk, err_msg=f"n={n} pass={p} rev={rev}")
                if vb:
                    np.testing.assert_array_equal(to_host(ov, vals.dtype), rv, err_msg=f"vals n={n} pass={p}")
        s.close()


def _size_ladder(P):
    small = [1, 2, 3, 63, 64, 65, 127, 255, 256, 257, 1000, 4095, 4096, 4097]
    around = [P - 1, P, P + 1, 2 * P - 1, 2 * P, 2 * P + 1, 3 * P + 1]
    ladder = list(range(P, 2 * P + 1, 509))
    big = [65536, 100003, (1 << 20) + 3]
    return sorted(set(small + around + ladder + big))


def test_keys_sort_parity_size_ladder(gpu, oracle, P):
    """Reference test shape: every partial-tile remainder class in [P, 2P] (OneSweepDispatcher.cuh:98-113), seed = size."""
    s = gpu.OneSweep((1 << 20) + 3)
    for n in _size_ladder(P):
        keys = oracle.init_random(n, n, 0)
        dk = to_dev(keys)
        s.sort(dk)
        s.check()
        np.testing.assert_array_equal(to_host(dk, np.uint32), oracle.std_sort(keys), err_msg=f"n={n}")
    s.close()


@pytest.mark.parametrize("kt", [0, 1, 2])
@pytest.mark.parametrize("order", [0, 1])
def test_key_types_and_orders(gpu, oracle, P, kt, order):
    """D3D12 SuperTestOneSweep matrix, keys (Tests.h:6-186): {asc,desc} x {uint,int,float}."""
    for n in (2, 1000, P + 7, 65536, 300001):
        keys = oracle.init_random(n, 5 + n, 0)
        out, _ = _gpu_sort(gpu, keys, kt, order)
        np.testing.assert_array_equal(out, oracle.std_sort(keys, kt, order), err_msg=f"n={n}")
        assert oracle.validate(out, kt, order) == 0


@pytest.mark.parametrize("vb", [4, 8])
@pytest.mark.parametrize("kt,order", [(0, 0), (0, 1), (1, 0), (2, 1)])
def test_pairs_parity_and_stability(gpu, oracle, P, vb, kt, order):
    """value = original index exposes stability; duplicates forced with entropy preset 4."""
    for n, andc in ((1, 0), (65, 3), (P, 0), (P + 1, 3), (2 * P + 3, 0), (250007, 3), ((1 << 20) + 3, 0)):
        keys = oracle.init_random(n, 11 + n, andc)
        vals = np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        ok, ov = _gpu_sort(gpu, keys, kt, order, vals)
        rk, rv = oracle.std_sort(keys, kt, order, vals)
        np.testing.assert_array_equal(ok, rk, err_msg=f"keys n={n}")
        np.testing.assert_array_equal(ov, rv, err_msg=f"values n={n}")


def test_pairs_reference_payload_convention(gpu, oracle, P):
    """The reference's own pairs check: payload := key, both must come out sorted (UtilityKernels.cuh:432-479)."""
    import torch
    n = 3 * P + 5
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    dv = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.init_random(dk, n, 0, dv)
    s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS, value_bytes=4)
    s.sort(dk, dv)
    s.check()
    assert gpu.validate(dk, dv) == 0
    np.testing.assert_array_equal(to_host(dk, np.uint32), to_host(dv, np.uint32))
    s.close()


@pytest.mark.parametrize("andc", [0, 1, 2, 3, 4])
def test_entropy_presets_u64_values(gpu, oracle, andc):
    """BASELINE config 5 at test size: (u32 key, u64 value) under the 5 Thearling-Smith presets."""
    n = (1 << 19) + 1
    keys = oracle.init_random(n, 10, andc)
    vals = np.arange(n, dtype=np.uint64) * np.uint64(0x100000001)
    ok, ov = _gpu_sort(gpu, keys, 0, 0, vals)
    rk, rv = oracle.std_sort(keys, 0, 0, vals)
    np.testing.assert_array_equal(ok, rk)
    np.testing.assert_array_equal(ov, rv)


def test_degenerate_distributions(gpu, oracle, P):
    n = 2 * P + 100
    for keys in (np.zeros(n, np.uint32), np.full(n, 0xFFFFFFFF, np.uint32), np.arange(n, dtype=np.uint32)[::-1].copy(),
                 (np.arange(n, dtype=np.uint32) % 3) << 24, np.arange(n, dtype=np.uint32)):
        vals = np.arange(n, dtype=np.uint32)
        ok, ov = _gpu_sort(gpu, keys, 0, 0, vals)
        rk, rv = oracle.std_sort(keys, 0, 0, vals)
        np.testing.assert_array_equal(ok, rk)
        np.testing.assert_array_equal(ov, rv)


def test_golden_vectors_on_gpu(gpu):
    g = np.load(GOLDEN)
    import torch
    for ci, row in enumerate(g["cases"]):
        n, seed, andc, kt, order, vb = (int(x) for x in row)
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.init_random(dk, seed, andc)
        torch.cuda.synchronize()
        keys = to_host(dk, np.uint32).copy()
        assert crc(keys) == int(g[f"c{ci}_in_crc"]), f"generator case {ci}"
        vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        ok, ov = _gpu_sort(gpu, keys, kt, order, vals)
        assert crc(ok) == int(g[f"c{ci}_out_crc"]), f"sort case {ci}"
        np.testing.assert_array_equal(ok[:16], g[f"c{ci}_out_head"])
        if vb:
            assert crc(ov) == int(g[f"c{ci}_vout_crc"]), f"values case {ci}"


def test_handle_reuse_and_caller_owned_alt(gpu, oracle, P):
    """One handle, many sizes, caller-provided temp buffers (Unity Sort signature, OneSweep.cs:297-323)."""
    import torch
    s = gpu.OneSweep(1 << 18)
    alt = torch.empty(1 << 18, dtype=torch.int32, device="cuda")
    for n in (1 << 18, 17, P + 3, 1 << 18, 1):
        keys = oracle.init_random(n, 1000 + n, 0)
        dk = to_dev(keys)
        s.sort(dk, alt_keys=alt)
        s.check()
        np.testing.assert_array_equal(to_host(dk, np.uint32), np.sort(keys))
    s.close()


def test_error_behaviour(gpu):
    import torch
    s = gpu.OneSweep(1000)
    k = torch.zeros(2000, dtype=torch.int32, device="cuda")
    with pytest.raises(gpu.GpuSortError) as e:
        s.sort(k, n=1001)
    assert e.value.status == 2  # GS_ERR_SIZE
    with pytest.raises(gpu.GpuSortError) as e:
        s.sort(k, n=0)
    assert e.value.status == 2
    with pytest.raises(gpu.GpuSortError) as e:
        s.sort(k[1:], n=10)  # misaligned (4-byte offset)
    assert e.value.status == 1  # GS_ERR_ARG
    with pytest.raises(ValueError):
        s.sort(k, k.clone(), n=10)  # values on a keys-only sorter
    s.close()


def test_dispatcher_quick(gpu):
    """OneSweepDispatcher::TestAllKeysOnly / TestAllPairs at CI size (quick ladder)."""
    lines = []
    d = gpu.OneSweepDispatcher(True, 1 << 20, out=lines.append)
    assert d.TestAllKeysOnly(quick=True), lines
    d2 = gpu.OneSweepDispatcher(False, 1 << 20, out=lines.append)
    assert d2.TestAllPairs(quick=True), lines
    assert any("All tests passed" in l for l in lines)


def test_every_compiled_shape(gpu, oracle):
    for t, k in ((512, 32), (512, 16), (256, 32), (1024, 16), (256, 16)):
        s = gpu.OneSweep(1 << 20)
        try:
            s.set_shape(t, k)
        except gpu.GpuSortError:
            s.close()
            continue
        for n in (t * k + 1, (1 << 20) - 5):
            keys = oracle.init_random(n, n, 0)
            dk = to_dev(keys)
            s.sort(dk)
            s.check()
            np.testing.assert_array_equal(to_host(dk, np.uint32), np.sort(keys), err_msg=f"{t}x{k} n={n}")
        s.close()


def test_lds_atomic_order_probe(gpu):
    """The returning-LDS-atomic ranking is only selected when this device probe finds no out-of-order lane."""
    import ctypes as C
    import torch
    from gpusorting_amd import _lib
    fails = C.c_uint64(123)
    st = _lib.load().gs_selftest_lds_atomic_order(200, 7, C.byref(fails), int(torch.cuda.current_stream().cuda_stream))
    assert st == 0 and fails.value == 0


@pytest.mark.parametrize("rank_mode", [0, 1])
@pytest.mark.parametrize("vb", [0, 4])
def test_both_ranking_paths(gpu, oracle, P, rank_mode, vb):
    """RANK 0 (64-lane ballot multi-split, the guaranteed path) and RANK 1 (LDS atomic) give identical results."""
    for n, andc in ((777, 0), (P + 5, 2), (3 * P + 1, 0), ((1 << 20) + 9, 4)):
        keys = oracle.init_random(n, 3 * n + 1, andc)
        vals = None if not vb else np.arange(n, dtype=np.uint32)
        s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, value_bytes=vb)
        s.set_rank_mode(rank_mode)
        dk = to_dev(keys)
        dv = None if not vb else to_dev(vals)
        s.sort(dk, dv)
        s.check()
        ref = oracle.std_sort(keys, vals=vals)
        rk, rv = (ref, None) if not vb else ref
        np.testing.assert_array_equal(to_host(dk, np.uint32), rk, err_msg=f"n={n} rank={rank_mode}")
        if vb:
            np.testing.assert_array_equal(to_host(dv, np.uint32), rv)
        s.close()


def test_chain_boundaries_and_skewed_chains(gpu, oracle, P):
    """Multi-chain specifics: digit groups of very different sizes, empty chains, chains that start mid-line."""
    rng = np.random.default_rng(5)
    n = 5 * P + 37
    cases = [
        (rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) & np.uint32(0x0F0F0F0F)),   # 16 digit values per byte
        (rng.integers(0, 3, n).astype(np.uint32) * np.uint32(0x01010101)),                            # 3 values, 13 empty chains
        np.where(rng.random(n) < 0.9, np.uint32(0x10101010), rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)),
        (np.arange(n, dtype=np.uint32) * np.uint32(2654435761)),
    ]
    for keys in cases:
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        for order in (0, 1):
            ok, ov = _gpu_sort(gpu, keys, 0, order, vals)
            rk, rv = oracle.std_sort(keys, 0, order, vals)
            np.testing.assert_array_equal(ok, rk)
            np.testing.assert_array_equal(ov, rv)


@pytest.mark.parametrize("pairs", [False, True])
def test_sharded_path_single_rank_nccl(gpu, oracle, pairs):
    """The multi-GPU pipeline (top-byte histogram -> all_gather -> splitters -> stable MSD partition ->
    all_to_all_single -> local sort) on ONE rank over RCCL, forced through the exchange path."""
    import torch
    import torch.distributed as dist
    from gpusorting_amd.sharded import ShardedOneSweep
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        n = (1 << 20) + 77
        keys = oracle.init_random(n, 99, 1)
        vals = np.arange(n, dtype=np.uint32) if pairs else None
        s = ShardedOneSweep(n, pairs=pairs, value_bytes=4, always_exchange=True)
        bk, bv, nb = s.sort(to_dev(keys), values=None if not pairs else to_dev(vals))
        s.engine.sorter.check()
        assert nb == n
        ref = oracle.std_sort(keys, vals=vals)
        rk, rv = (ref, None) if not pairs else ref
        np.testing.assert_array_equal(to_host(bk, np.uint32), rk)
        if pairs:
            np.testing.assert_array_equal(to_host(bv, np.uint32), rv)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("pairs,andc,kt", [(False, 0, 0), (True, 0, 2), (False, 3, 1), (True, 3, 0)])
def test_sharded_path_single_rank_bucket_landed_in_the_alternate_buffer(gpu, oracle, monkeypatch, pairs, andc, kt):
    """Round 6: a bucket that is offered the two-level plan is landed in the local sort's ALTERNATE buffer (here by the RCCL transport's
    own-bucket copy: one rank, exchange forced) and the local sort starts at the plan's second pass — the split pass was its first.
    The sorter is offered the plan from 2^20 keys (gs_mgpu_options::sorter: plan 2) so that the path runs at a size the oracle checks
    in a moment; uniform keys run the plan, entropy preset 4 voids it on the device (hy_void_copy_kernel, four LSD passes); typed keys."""
    import torch
    import torch.distributed as dist
    from gpusorting_amd.sharded import ShardedOneSweep
    monkeypatch.setenv("GPUSORT_PLAN", "2")
    monkeypatch.setenv("GPUSORT_POS_MIN_LOG2", "20")
    monkeypatch.setenv("GPUSORT_MID_PATH", "0")   # (below 2^23 keys the two-launch route would take the bucket)
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        n = (5 << 20) + 1234
        keys = oracle.init_random(n, 4321 + andc, andc)
        vals = np.arange(n, dtype=np.uint32) if pairs else None
        s = ShardedOneSweep(n, pairs=pairs, value_bytes=4, always_exchange=True, key_type=kt)
        for rep in range(2):   # (twice on one context: the buffers' roles swap back)
            bk, bv, nb = s.sort(to_dev(keys), values=None if not pairs else to_dev(vals))
            s.check()
            assert nb == n and s.last_bin_major
            assert s.engine.sorter.last_plan()["two_level"] == (andc == 0)
            ref = oracle.std_sort(keys, kt, 0, vals)
            rk, rv = (ref, None) if not pairs else ref
            np.testing.assert_array_equal(to_host(bk, np.uint32), rk)
            if pairs:
                np.testing.assert_array_equal(to_host(bv, np.uint32), rv)
        s.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("small_path", [True, False])
@pytest.mark.parametrize("vb,kt,order", [(0, 0, 0), (0, 2, 1), (4, 1, 1), (8, 0, 0), (4, 0, 0)])
def test_single_tile_path_and_tiled_path_agree(gpu, oracle, small_path, vb, kt, order):
    """Small n takes the one-launch single-tile kernel (8192 slots for every mode, 16384 for keys-only and
    4-byte values, 32768 for keys-only); with it switched off the same sizes go through histogram + scan +
    4 passes.  Both must equal the oracle (stability: value = index)."""
    for n, andc in ((1, 0), (2, 0), (63, 1), (64, 0), (65, 4), (1000, 0), (4097, 2), (8191, 0), (8192, 3), (8193, 0),
                    (12345, 1), (16384, 0), (16385, 4), (30000, 0), (32768, 2), (32769, 0)):
        keys = oracle.init_random(n, 7 * n + 3, andc)
        vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        s = gpu.OneSweep(n, order, kt, gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, vb)
        s.set_small_path(small_path)
        s.set_mid_path(small_path)
        for rank in (0, 1):
            s.set_rank_mode(rank)
            dk = to_dev(keys)
            dv = None if not vb else to_dev(vals)
            s.sort(dk, dv)
            s.check()
            ref = oracle.std_sort(keys, kt, order, vals)
            rk, rv = (ref, None) if not vb else ref
            np.testing.assert_array_equal(to_host(dk, np.uint32), rk, err_msg=f"n={n} small={small_path} rank={rank}")
            if vb:
                np.testing.assert_array_equal(to_host(dv, vals.dtype), rv, err_msg=f"values n={n} small={small_path}")
        s.close()


@pytest.mark.parametrize("n,vb", [(5000, 0), ((1 << 20) + 3, 0), ((1 << 18) + 1, 4)])
def test_sort_is_hip_graph_capturable(gpu, oracle, n, vb):
    """A sort is six kernels on the caller's stream and nothing synchronous, so it can be
    captured once into a HIP graph and replayed on new data in the same buffers."""
    import torch
    s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, value_bytes=vb)
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    dv = torch.empty(n, dtype=torch.int32, device="cuda") if vb else None
    alt = torch.empty(n, dtype=torch.int32, device="cuda")
    valt = torch.empty(n, dtype=torch.int32, device="cuda") if vb else None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s.sort(dk, dv, alt_keys=alt, alt_values=valt)  # warm-up outside capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        s.sort(dk, dv, alt_keys=alt, alt_values=valt)
    for seed in (3, 4):
        keys = oracle.init_random(n, seed, 1)
        vals = np.arange(n, dtype=np.uint32)
        dk.copy_(torch.from_numpy(keys.view(np.int32)))
        if vb:
            dv.copy_(torch.from_numpy(vals.view(np.int32)))
        graph.replay()
        torch.cuda.synchronize()
        ref = oracle.std_sort(keys, vals=vals if vb else None)
        rk, rv = (ref, None) if not vb else ref
        np.testing.assert_array_equal(to_host(dk, np.uint32), rk)
        if vb:
            np.testing.assert_array_equal(to_host(dv, np.uint32), rv)
    s.check()
    s.close()


@pytest.mark.parametrize("key_dtype,kt", [("uint32", 0), ("int32", 1), ("float32", 2)])
@pytest.mark.parametrize("val_dtype", ["int32", "float32"])
@pytest.mark.parametrize("order", [0, 1])
def test_supertest_matrix_native_dtypes(gpu, oracle, key_dtype, kt, val_dtype, order):
    """D3D12 SuperTestOneSweep (Tests.h:6-186): {asc,desc} x {uint,int,float} keys x {uint,int,float}
    payloads, on tensors of the NATIVE dtypes; payloads are bit-copied whatever their type."""
    import torch
    n = 70001
    raw = oracle.init_random(n, 17 + kt, 0)
    if kt == 2:  # make them honest floats: finite, both signs, duplicates
        raw = (np.random.default_rng(3).standard_normal(n).astype(np.float32) * 1000).round(1).view(np.uint32)
    payload = (np.arange(n, dtype=np.float32) * 0.5) if val_dtype == "float32" else np.arange(n, dtype=np.int32) - 5
    tk = torch.from_numpy(raw.view(np.int32).copy()).cuda()
    if key_dtype == "float32":
        tk = tk.view(torch.float32)
    elif key_dtype == "uint32" and hasattr(torch, "uint32"):
        tk = tk.view(torch.uint32)
    tv = torch.from_numpy(payload.copy()).cuda()
    s = gpu.OneSweep(n, order, kt, gpu.MODE_PAIRS, 4)
    s.sort(tk, tv)
    s.check()
    rk, rv = oracle.std_sort(raw, kt, order, payload.view(np.uint32))
    np.testing.assert_array_equal(tk.view(torch.int32).cpu().numpy().view(np.uint32), rk)
    np.testing.assert_array_equal(tv.view(torch.int32).cpu().numpy().view(np.uint32), rv)
    if kt == 2 and order == 0:
        f = tk.view(torch.float32).cpu().numpy()
        assert np.all(f[:-1] <= f[1:])
    s.close()


_MASKS = [0x0000FFFF, 0x00FFFFFF, 0x000000FF, 0xFF00FF00, 0xFFFF0000, 0x00FF0000, 0x12000000, 0x00000000]


@pytest.mark.parametrize("skip", [True, False])
@pytest.mark.parametrize("vb,kt,order", [(0, 0, 0), (0, 0, 1), (4, 0, 0), (4, 0, 1), (8, 1, 1), (0, 2, 0), (4, 2, 1)])
def test_identity_passes_dropped_on_device(gpu, oracle, P, skip, vb, kt, order):
    """passes whose digit is the same for every key are dropped (in pairs, decided by the Scan
    kernel) — results must be what the full four passes give, including the descending reversal of values
    of equal keys, which needs a pass to carry it even when no digit varies."""
    n = 5 * P + 77
    base = oracle.init_random(n, 4242, 0)
    for mask in _MASKS:
        keys = (base & np.uint32(mask)) | np.uint32(0x5A000000 & ~mask)   # constant bytes are not all zero
        vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        s = gpu.OneSweep(n, order, kt, gpu.MODE_KEYS_ONLY if not vb else gpu.MODE_PAIRS, vb)
        s.set_skip_passes(skip)
        dk = to_dev(keys)
        dv = None if vals is None else to_dev(vals)
        s.sort(dk, dv)
        s.check()
        if vals is None:
            np.testing.assert_array_equal(to_host(dk, np.uint32), oracle.std_sort(keys, kt, order), err_msg=hex(mask))
        else:
            rk, rv = oracle.std_sort(keys, kt, order, vals)
            np.testing.assert_array_equal(to_host(dk, np.uint32), rk, err_msg=hex(mask))
            np.testing.assert_array_equal(to_host(dv, vals.dtype), rv, err_msg=hex(mask))
        s.close()


def test_dropped_passes_cost_nothing(gpu, routing):
    """16-bit keys: passes 2 and 3 must be launches of workgroups that exit at once — on the position-chain plan too (its passes
    learn their digit counts only from the pass before; the OR / AND of all keys, accumulated by the histogram kernel, tell the
    Scan kernel which bytes are constant)."""
    import torch
    n = 1 << 24
    k = torch.randint(0, 1 << 16, (n,), dtype=torch.int32, device="cuda")
    s = gpu.OneSweep(n)
    s.set_profiling(True)
    best = None
    for _ in range(3):
        kk = k.clone()
        s.sort(kk)
        torch.cuda.synchronize()
        p = s.get_profile()
        best = p if best is None or p["total"] < best["total"] else best
    assert bool((kk[1:] >= kk[:-1]).all().item())
    real = min(best["pass0"], best["pass1"])
    assert max(best["pass2"], best["pass3"]) < 0.5 * real, best
    s.close()


def _heavy_inputs(oracle, n):
    rng = np.random.default_rng(99)
    u = oracle.init_random(n, 777, 0)
    yield "and-skew p=1/8", oracle.init_random(n, 31, 2)
    yield "and-skew p=1/32", oracle.init_random(n, 33, 4)
    # every byte is 0x35 with probability 1/2, else uniform: the heavy value sits in the middle of its digit
    # group, with neighbours below and above it in every pass
    k = u.copy()
    for b in range(4):
        pick = rng.random(n) < 0.5
        k = np.where(pick, (k & np.uint32(~(0xFF << (8 * b)) & 0xFFFFFFFF)) | np.uint32(0x35 << (8 * b)), k)
    yield "0x35 half of every byte", k.astype(np.uint32)
    yield "constant middle byte", ((u & np.uint32(0xFF00FFFF)) | np.uint32(0x00C30000)).astype(np.uint32)
    yield "90% one key", np.where(rng.random(n) < 0.9, np.uint32(0x80402010), u).astype(np.uint32)
    yield "heavy value 255", (u | np.where(rng.random(n) < 0.6, np.uint32(0x0000FF00), np.uint32(0))).astype(np.uint32)


@pytest.mark.parametrize("vb,kt,order", [(0, 0, 0), (4, 0, 0), (8, 0, 1), (0, 1, 1), (4, 2, 0)])
def test_skewed_keys_take_position_chains(gpu, oracle, vb, kt, order, monkeypatch, routing):
    """Skewed keys (uneven digit groups): the histogram kernel notices, the Scan kernel plans every pass on position
    chains, each pass counts the next one's digit per output segment while it scatters.  The library allows that plan
    from 2^25 keys up; GPUSORT_POS_MIN_LOG2 lowers the threshold for this test (keys-only sorts and pairs of both value widths)."""
    if routing != "position-chains":
        monkeypatch.setenv("GPUSORT_POS_MIN_LOG2", "22")
    n = (1 << 22) + 54321
    for name, keys in _heavy_inputs(oracle, n):
        vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        ok, ov = _gpu_sort(gpu, keys, kt, order, vals)
        if vals is None:
            np.testing.assert_array_equal(ok, oracle.std_sort(keys, kt, order), err_msg=name)
        else:
            rk, rv = oracle.std_sort(keys, kt, order, vals)
            np.testing.assert_array_equal(ok, rk, err_msg=name)
            np.testing.assert_array_equal(ov, rv, err_msg=name)


@pytest.mark.parametrize("kt,order", [(0, 0), (1, 1), (2, 0)])
def test_mid_size_first_pass_on_the_larger_tile(gpu, oracle, kt, order):
    """2^22 < n <= 2^25 keys-only on the general path: the first pass runs on 16 384-key tiles over tile-aligned position
    segments, the later passes on 8192-key tiles (round 3) — two tile shapes in one plan.  Exact, and the scan state adds up
    (every pass accounts for n keys with ITS tile size: _assert_scan_state in _gpu_sort)."""
    for n in ((1 << 22) + 1, (1 << 23) + 777, 3 * (1 << 22) - 5):
        keys = oracle.init_random(n, 23 + kt, 0)
        ok, _ = _gpu_sort(gpu, keys, kt, order)
        np.testing.assert_array_equal(ok, oracle.std_sort(keys, kt, order), err_msg=f"n={n} kt={kt} order={order}")


def _fuzz_keys(rng, oracle, n):
    kind = int(rng.integers(0, 8))
    u = oracle.init_random(n, int(rng.integers(1, 1 << 30)), 0)
    if kind == 0:
        return u
    if kind == 1:
        return oracle.init_random(n, int(rng.integers(1, 1 << 30)), int(rng.integers(1, 5)))
    if kind == 2:  # few distinct values
        pool = rng.integers(0, 1 << 32, size=int(rng.integers(1, 9)), dtype=np.uint64).astype(np.uint32)
        return pool[rng.integers(0, pool.size, size=n)]
    if kind == 3:  # sorted / reversed / nearly sorted
        s = np.sort(u)
        return s if rng.random() < 0.5 else s[::-1].copy()
    if kind == 4:  # random constant bytes
        mask = np.uint32(sum(0xFF << (8 * b) for b in range(4) if rng.random() < 0.5))
        return (u & mask) | (np.uint32(rng.integers(0, 1 << 32, dtype=np.uint64)) & ~mask)
    if kind == 5:  # one heavy value per byte with random weight
        k = u.copy()
        for b in range(4):
            pick = rng.random(n) < rng.random()
            hv = np.uint32(int(rng.integers(0, 256)) << (8 * b))
            k = np.where(pick, (k & np.uint32(~(0xFF << (8 * b)) & 0xFFFFFFFF)) | hv, k)
        return k.astype(np.uint32)
    if kind == 6:  # small range
        return (u % np.uint32(int(rng.integers(1, 70000)))).astype(np.uint32)
    return (u >> np.uint32(int(rng.integers(0, 31)))).astype(np.uint32)


def test_fuzz_against_oracle(gpu, oracle, monkeypatch, routing):
    """Seeded random sweep over sizes (1 .. 6M, so every path: single tile, 8192-key tiles, position-chain plan —
    its size threshold lowered), distributions, key types, orders and value widths;
    every case bit-exact against the oracle."""
    monkeypatch.setenv("GPUSORT_POS_MIN_LOG2", "21")
    rng = np.random.default_rng(int(os.environ.get("GPUSORT_FUZZ_SEED", "20260925")))
    for case in range(int(os.environ.get("GPUSORT_FUZZ_CASES", "48"))):  # longer hunts: set the two variables
        top = (40000, 300000, 6 << 20)[case % 3]
        n = int(rng.integers(1, top))
        keys = np.ascontiguousarray(_fuzz_keys(rng, oracle, n), dtype=np.uint32)
        kt, order, vb = int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.choice([0, 0, 4, 8]))
        vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        ok, ov = _gpu_sort(gpu, keys, kt, order, vals)
        ref = oracle.std_sort(keys, kt, order, vals)
        rk, rv = (ref, None) if vals is None else ref
        np.testing.assert_array_equal(ok, rk, err_msg=f"case {case}: n={n} kt={kt} order={order} vb={vb}")
        if vals is not None:
            np.testing.assert_array_equal(ov, rv, err_msg=f"case {case} values: n={n} kt={kt} order={order} vb={vb}")


@pytest.mark.parametrize("pairs", [False, True])
def test_unity_light_test_matrix(gpu, oracle, pairs):
    """Unity's light tests (GPUSortingUnity/Tests/TestBase.cs:238-264 and the pairs twin): every power of two
    from 2 to 65536 x {ascending, descending} x {uint, int, float} keys, seed = size — 96 sorts, each bit-exact."""
    passed = 0
    for lg in range(1, 17):
        n = 1 << lg
        keys = oracle.init_random(n, n, 0)
        vals = np.arange(n, dtype=np.uint32) if pairs else None
        for kt in (0, 1, 2):
            for order in (0, 1):
                ok, ov = _gpu_sort(gpu, keys, kt, order, vals)
                ref = oracle.std_sort(keys, kt, order, vals)
                rk, rv = (ref, None) if vals is None else ref
                np.testing.assert_array_equal(ok, rk, err_msg=f"n={n} kt={kt} order={order}")
                if pairs:
                    np.testing.assert_array_equal(ov, rv, err_msg=f"values n={n} kt={kt} order={order}")
                passed += 1
    assert passed == 96


_REF_SORT_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_onesweep.so")


@pytest.mark.skipif(not os.path.exists(_REF_SORT_LIB), reason="oracle/_ref is built only where /root/reference exists")
def test_gpu_equals_reference_kernels_run_live(gpu):
    """Head to head on this box: the reference's own OneSweep kernels (oracle/_ref: OneSweep.cu under the SIMT
    emulator, on the CPU) and the HIP path sort the SAME device-generated input; keys and payloads must be equal."""
    import ctypes as C
    import torch
    ref = C.CDLL(_REF_SORT_LIB)
    ref.ref_onesweep_sort_keys.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    ref.ref_onesweep_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(77)
    for n in [7680, 15361] + [int(rng.integers(2, 40000)) for _ in range(5)]:
        andc = int(rng.integers(0, 5))
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.init_random(dk, n + 1, andc)
        dv = torch.arange(n, dtype=torch.int32, device="cuda")
        hk = to_host(dk, np.uint32).copy()
        hv = np.arange(n, dtype=np.uint32)
        ref.ref_onesweep_sort_pairs(hk.ctypes.data, hv.ctypes.data, n, None, None, None)   # the reference, on the CPU
        s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS, value_bytes=4)
        s.sort(dk, dv)                                                                       # this library, on the GPU
        s.check()
        np.testing.assert_array_equal(to_host(dk, np.uint32), hk, err_msg=f"keys n={n} preset={andc + 1}")
        np.testing.assert_array_equal(to_host(dv, np.uint32), hv, err_msg=f"payloads n={n} preset={andc + 1}")
        s.close()


def test_two_handles_on_two_streams_concurrently(gpu, oracle):
    """One handle per stream (include/gpusort.h): two sorts in flight at once share nothing — every piece of scan
    state lives in the handle's slab."""
    import torch
    n1, n2 = (1 << 22) + 11, (1 << 21) + 7
    k1, k2 = oracle.init_random(n1, 101, 0), oracle.init_random(n2, 202, 3)
    v2 = np.arange(n2, dtype=np.uint32)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    h1 = gpu.OneSweep(n1)
    h2 = gpu.OneSweep(n2, order=gpu.ORDER_DESCENDING, mode=gpu.MODE_PAIRS, value_bytes=4)
    d1, d2, dv2 = to_dev(k1), to_dev(k2), to_dev(v2)
    torch.cuda.synchronize()
    for rep in range(3):   # sorted input again and again: results must not change
        with torch.cuda.stream(s1):
            h1.sort(d1)
        with torch.cuda.stream(s2):
            h2.sort(d2, dv2)
    s1.synchronize()
    s2.synchronize()
    np.testing.assert_array_equal(to_host(d1, np.uint32), oracle.std_sort(k1))
    rk, rv = oracle.std_sort(k2, 0, 1, v2)
    np.testing.assert_array_equal(to_host(d2, np.uint32), rk)
    # descending re-sorts of an already descending array reverse equal keys each time: 3 sorts = 1 reversal
    np.testing.assert_array_equal(to_host(dv2, np.uint32), rv)
    h1.close()
    h2.close()


def test_tensor_convenience_layer(gpu, oracle):
    """gpusorting_amd.sort / sort_ / argsort: dtype -> key type, cached handles, growing sizes."""
    import torch
    for n in (1, 5, 40000, 70000, (1 << 20) + 1, 1000):          # the cached handle grows and is reused
        bits = oracle.init_random(n, n + 9, 1)
        for dtype, kt in ((torch.int32, 1), (torch.float32, 2)):
            t = torch.from_numpy(bits.view(np.int32)).cuda().view(dtype)
            for desc in (False, True):
                out = gpu.sort(t, descending=desc)
                np.testing.assert_array_equal(out.view(torch.int32).cpu().numpy().view(np.uint32),
                                              oracle.std_sort(bits, kt, int(desc)), err_msg=f"n={n} {dtype} desc={desc}")
        t = torch.from_numpy(bits.view(np.int32)).cuda()
        vals = torch.arange(n, dtype=torch.int64, device="cuda")
        k2, v2 = gpu.sort(t, vals, unsigned=True)
        rk, rv = oracle.std_sort(bits, 0, 0, np.arange(n, dtype=np.uint64))
        np.testing.assert_array_equal(k2.cpu().numpy().view(np.uint32), rk)
        np.testing.assert_array_equal(v2.cpu().numpy().view(np.uint64), rv)
        perm = gpu.argsort(t, unsigned=True)
        np.testing.assert_array_equal(perm.cpu().numpy().astype(np.uint64), rv)   # stable: same permutation
    with pytest.raises(TypeError):
        gpu.sort(torch.zeros(8, dtype=torch.int64, device="cuda"))


def test_two_host_threads_each_with_its_own_handle(gpu, oracle):
    """One handle per thread (include/gpusort.h): creation (incl. the once-per-device LDS probe) and sorting from
    two host threads at the same time; ctypes releases the GIL around every call."""
    import threading
    import torch
    errors = []

    def work(seed, n, pairs):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            keys = oracle.init_random(n, seed, 1)
            vals = np.arange(n, dtype=np.uint32) if pairs else None
            with torch.cuda.stream(stream):
                for _ in range(3):
                    s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if pairs else gpu.MODE_KEYS_ONLY, value_bytes=4 if pairs else 0)
                    dk = to_dev(keys)
                    dv = None if vals is None else to_dev(vals)
                    s.sort(dk, dv)
                    s.check(stream)
                    ref = oracle.std_sort(keys, 0, 0, vals)
                    rk, rv = (ref, None) if vals is None else ref
                    np.testing.assert_array_equal(to_host(dk, np.uint32), rk)
                    if pairs:
                        np.testing.assert_array_equal(to_host(dv, np.uint32), rv)
                    s.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(31, 300007, False)), threading.Thread(target=work, args=(32, (1 << 20) + 5, True))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_scan_state_invariants(gpu, oracle):
    """The slab-state checker itself: after tiled sorts of several shapes the report is clean and says which
    passes ran (16-bit keys: two identity passes dropped); after a single-tile sort it is all zero; after a
    stand-alone pass and after the histogram-only entry the histogram region is zero again."""
    import torch
    for n, andc, vb, mask in [(300000, 0, 0, 0xFFFFFFFF), (2500000, 0, 4, 0xFFFFFFFF), (1 << 22, 4, 0, 0xFFFFFFFF),
                              (700001, 0, 8, 0x0000FFFF), (5000, 0, 0, 0xFFFFFFFF)]:
        keys = oracle.init_random(n, 5 + andc, andc) & np.uint32(mask)
        s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, value_bytes=vb)
        s.set_mid_path(False)  # the scan state of the general pipeline is what this test looks at
        dk = to_dev(keys)
        dv = None if not vb else torch.arange(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
        s.sort(dk, dv)
        s.check()
        r = s.check_state()
        assert (r["rows_not_inclusive"], r["rows_not_monotone"], r["chains_short_of_tickets"], r["hist_words_nonzero"]) == (0, 0, 0, 0), r
        if n <= 8192:
            assert r["keys_per_pass"] == [0, 0, 0, 0]          # single-tile path: no scan state
        elif mask == 0x0000FFFF:
            assert r["keys_per_pass"] == [n, n, 0, 0], r        # bytes 2 and 3 are constant: dropped as a pair
        else:
            assert r["keys_per_pass"] == [n, n, n, n], r
        out = torch.empty_like(dk)
        s.digit_pass(dk, out, 1, values_in=dv, values_out=None if dv is None else torch.empty_like(dv))
        r = s.check_state()
        assert r["hist_words_nonzero"] == 0 and r["rows_not_inclusive"] == 0 and r["keys_per_pass"][0] == n, r
        s.global_histogram(dk)
        assert s.check_state()["hist_words_nonzero"] == 0
        s.close()


def test_wave_primitives_against_the_host(gpu):
    """the wave-level primitives (the wave64 counterparts of GPUSortingCUDA/Utils.cuh:22-126 and of the ballot
    multi-split of OneSweep.cu:207-253) checked on their own, lane by lane, against numpy: inclusive scans (shuffle form and DPP form),
    reduction, the 64-bit ballot, mbcnt as the lanemask_lt popcount, and the rank of a lane among the lanes holding its digit."""
    import torch
    from gpusorting_amd import _lib
    waves, seed = 64, 20260930
    out = torch.zeros(waves * 512, dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().gs_selftest_wave_primitives(seed, waves, out.data_ptr(), int(torch.cuda.current_stream().cuda_stream)), "selftest")
    torch.cuda.synchronize()
    r = out.cpu().numpy().view(np.uint32).reshape(waves, 8, 64)
    i = np.arange(waves * 64, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    h = ((np.uint64(seed) ^ ((i * np.uint64(2654435761)) & M)) * np.uint64(2246822519)) & M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(3266489917)) & M
    x = (h ^ (h >> np.uint64(16))).astype(np.uint32).reshape(waves, 64)
    np.testing.assert_array_equal(r[:, 0], x)
    lo = (x & 0xFFFF).astype(np.uint64)
    incl = np.cumsum(lo, axis=1).astype(np.uint32)
    np.testing.assert_array_equal(r[:, 1], incl)                       # wave_inclusive_scan (shuffles)
    np.testing.assert_array_equal(r[:, 2], incl)                       # wave_inclusive_scan_dpp
    np.testing.assert_array_equal(r[:, 3], np.repeat(incl[:, 63:64], 64, axis=1))   # wave_reduce_sum, lane 0's value broadcast
    bit = (x & 1).astype(np.uint64)
    ballot = (bit << np.arange(64, dtype=np.uint64)[None, :]).sum(axis=1)
    np.testing.assert_array_equal(r[:, 4], np.repeat((ballot & M).astype(np.uint32)[:, None], 64, axis=1))
    np.testing.assert_array_equal(r[:, 5], np.repeat((ballot >> np.uint64(32)).astype(np.uint32)[:, None], 64, axis=1))
    below = (np.cumsum(bit, axis=1) - bit).astype(np.uint32)           # set lanes below this lane
    np.testing.assert_array_equal(r[:, 6], below)
    d = x & 0xFF
    rank = np.zeros_like(d)
    for l in range(64):
        rank[:, l] = (d[:, :l] == d[:, l:l + 1]).sum(axis=1)
    np.testing.assert_array_equal(r[:, 7], rank)                       # the multi-split's rank among equal digits
