"""GPU tests at BASELINE.json's full sizes: bit-exact parity at 2^28 against the CPU
oracle for configs[1] (keys), configs[2] (u32 values) and configs[4] (u64 values, entropy
presets 1 and 5) with value = original index (the only payload that exposes a stability
violation), one exact case through the ballot ranking path at 2^26, plus size-independent
properties (sortedness, permutation-invariant checksums, digit histograms preserved,
idempotence) and the maximum size 2^30 - 1.
Reference sizes: GPUSortingCUDA/Sort/OneSweepDispatcher.cuh:116-128,166-185 (2^26, 2^27, 2^28)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _checksums(t):
    import torch
    v = t.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return int(v.sum().item()), int((v * v % 1000003).sum().item())


def test_2pow26_exact_vs_oracle(gpu, oracle):
    import torch
    n = 1 << 26
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.init_random(dk, 26, 0)
    torch.cuda.synchronize()
    keys = dk.cpu().numpy().view(np.uint32).copy()
    s = gpu.OneSweep(n)
    s.sort(dk)
    s.check()
    ref = oracle.std_sort_parallel(keys, oracle.hardware_threads())
    np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), ref)
    s.close()


@pytest.mark.parametrize("pairs", [False, True])
def test_2pow28_properties(gpu, pairs):
    """configs[1]/[2]: 2^28 keys (pairs: u32 values = key, the reference's own convention)."""
    import torch
    n = 1 << 28
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    dv = torch.empty(n, dtype=torch.int32, device="cuda") if pairs else None
    gpu.init_random(dk, 28, 0, dv)
    before = _checksums(dk)
    s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if pairs else gpu.MODE_KEYS_ONLY, value_bytes=4 if pairs else 0)
    h_before = s.global_histogram(dk)
    s.sort(dk, dv)
    s.check()
    assert gpu.validate(dk, dv) == 0                      # the reference's pass criterion
    assert _checksums(dk) == before                        # permutation-invariant checksums
    np.testing.assert_array_equal(s.global_histogram(dk), h_before)  # every digit histogram preserved
    if pairs:
        assert bool((dk == dv).all().item())               # payload travelled with its key
    # idempotence: sorting sorted data changes nothing
    again = dk.clone()
    s.sort(again, None if not pairs else dv.clone())
    s.check()
    assert bool((again == dk).all().item())
    s.close()


@pytest.mark.parametrize("andc", [0, 4])
def test_maximum_size_2pow30_minus_1(gpu, andc):
    """Largest n the API accepts (30-bit tile-descriptor payload): 4 GiB of keys, uniform and at entropy preset 5
    (the position-chain plan with the largest possible counts).  Properties only: the reference's own pass criterion
    (no inversion) and all four digit histograms preserved."""
    import torch
    n = (1 << 30) - 1
    dk = torch.empty(n + 1, dtype=torch.int32, device="cuda")[:n]
    gpu.init_random(dk, 30, andc, n=n)
    s = gpu.OneSweep(n)
    h_before = s.global_histogram(dk, n)
    assert int(h_before[0].sum()) == n
    s.sort(dk, n=n)
    s.check()
    assert s.last_plan()["two_level"] == (andc == 0)     # uniform: the two-level plan's largest class (24 576-key buckets); preset 5: the LSD passes
    assert gpu.validate(dk, n=n) == 0
    np.testing.assert_array_equal(s.global_histogram(dk, n), h_before)
    lo = dk[:4].cpu().numpy().view(np.uint32)
    hi = dk[n - 4:n].cpu().numpy().view(np.uint32)
    assert lo[0] <= lo[-1] <= hi[0] <= hi[-1]
    s.close()
    with pytest.raises(gpu.GpuSortError):
        gpu.OneSweep(1 << 30)  # one past the limit: GS_ERR_SIZE


@pytest.mark.parametrize("kt,order,vb", [(1, 1, 4), (2, 0, 8), (2, 1, 0)])
def test_2pow24_typed_keys_exact(gpu, oracle, kt, order, vb):
    """Typed keys / descending / values at a multi-thousand-tile size, exact against the oracle."""
    import torch
    n = (1 << 24) + 12345
    keys = oracle.init_random(n, 24 + kt, 0)
    vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
    s = gpu.OneSweep(n, order, kt, gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, vb)
    dk = torch.from_numpy(keys.view(np.int32)).cuda()
    dv = None if not vb else torch.from_numpy(vals.view(np.int32 if vb == 4 else np.int64)).cuda()
    s.sort(dk, dv)
    s.check()
    ref = oracle.std_sort(keys, kt, order, vals)
    rk, rv = (ref, None) if not vb else ref
    np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), rk)
    if vb:
        np.testing.assert_array_equal(dv.cpu().numpy().view(vals.dtype), rv)
    s.close()


@pytest.mark.parametrize("andc,pairs", [(4, False), (3, False), (2, False), (4, True)])
def test_2pow28_low_entropy_properties(gpu, andc, pairs):
    """BASELINE configs[4]'s entropy presets at full size (presets 3-5): skewed ranking, and for keys-only
    sorts on the position-chain plan (12 288-key counting tiles, 16 384-key last pass).  Properties: no inversion, permutation-invariant
    checksums, all digit histograms preserved, payload travelled with its key."""
    import torch
    n = 1 << 28
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    dv = torch.empty(n, dtype=torch.int64, device="cuda") if pairs else None
    gpu.init_random(dk, 28 + andc, andc, dv)
    before = _checksums(dk)
    s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if pairs else gpu.MODE_KEYS_ONLY, value_bytes=8 if pairs else 0)
    h_before = s.global_histogram(dk)
    s.sort(dk, dv)
    s.check()
    assert gpu.validate(dk) == 0
    assert _checksums(dk) == before
    np.testing.assert_array_equal(s.global_histogram(dk), h_before)
    if pairs:  # the generator sets value = zero-extended key (UtilityKernels.cuh:157-168)
        assert bool(((dk.to(torch.int64) & 0xFFFFFFFF) == dv).all().item())
    s.close()


# ---- bit-exact at the headline size ------------------------------------------------------------
def _exact_case(gpu, oracle, log2n, andc, vb, rank_mode=None, order=0, kt=0, plan=None, n=None, two_level=None):
    """Sort n = 2^log2n (or `n`) generator keys (seed = log2n as the reference's big sizes, OneSweepDispatcher.cuh:116-128)
    with value = original index and compare keys AND values element for element with the oracle's stable
    order (descending: its exact reverse, SortCommon.hlsl:594-597,645-656; typed keys: :134-154).  Keys-only sorts of typed or
    descending keys are compared through the oracle's permutation as well.  The comparison itself runs on the GPU (a gather and
    two equality reductions: plumbing).  two_level: what gs_onesweep_last_plan must report for the sort."""
    import torch
    if n is None:
        n = 1 << log2n
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.init_random(dk, log2n + 100 * andc, andc)
    torch.cuda.synchronize()
    keys = dk.cpu().numpy().view(np.uint32)
    orig = dk.clone()
    dv = None
    if vb:
        dv = torch.arange(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
    s = gpu.OneSweep(n, order, kt, gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, vb)
    if rank_mode is not None:
        s.set_rank_mode(rank_mode)
    if plan is not None:
        s.set_plan(plan)
    s.sort(dk, dv)
    s.check()
    if two_level is not None:
        assert s.last_plan()["two_level"] == two_level, s.last_plan()
    if vb or kt or order:
        perm = oracle.sort_permutation_parallel(keys, kt, order)          # stable order by key (desc: its reverse)
        dperm = torch.from_numpy(perm.view(np.int32)).cuda()
        if vb:
            assert bool((dv.to(torch.int32) == dperm).all().item()), "payload order differs from the stable sort"
            if vb == 8:
                assert bool((dv >> 32 == 0).all().item())
        idx = dperm.to(torch.int64) & 0xFFFFFFFF
        del dperm
        expect = orig[idx]
        assert bool((dk == expect).all().item()), "sorted keys differ from the oracle"
    else:
        ref = oracle.std_sort_parallel(keys, oracle.hardware_threads())
        assert bool((dk == torch.from_numpy(ref.view(np.int32)).cuda()).all().item()), "sorted keys differ from the oracle"
    s.close()


# ---- typed keys / descending order on the two-level plan at the sizes the DEFAULT routing uses it ----
# The plan's descending rule: pass B writes mirrored, the bucket-local sort reads and writes mirrored (hybrid_kernels.hpp); its key
# transforms run at every load / store as in the LSD passes.  Bucket-sort classes by n: <= 2^27 (256 x 12), <= 2^28 (512 x 12),
# <= 2^29 (1024 x 12), above (1024 x 24).  Reference rule: GPUSortingD3D12/Shaders/SortCommon.hlsl:134-154,594-597,645-656; its
# test matrix: GPUSortingD3D12/Tests.h:6-186.
@pytest.mark.parametrize("n,kt,order,vb", [
    ((1 << 27) + 1, 1, 1, 0),   # class 1 at its lower border: int32 keys, descending
    (1 << 28, 2, 1, 0),         # class 1, the headline size: float keys, descending
    (1 << 28, 0, 1, 4),         # class 1: (u32, u32) pairs descending, value = index
    (1 << 27, 1, 1, 8),         # class 0 at its upper border: (i32, u64) pairs descending, value = index
    (3 << 27, 0, 1, 0),         # class 2: descending
    (3 << 27, 2, 0, 4),         # class 2: float keys with u32 values, ascending
    ((1 << 28) + 12345, 1, 0, 8),  # class 2 at its lower border, ragged: (i32, u64) pairs ascending
    ((1 << 29) + 77, 1, 1, 0),     # class 3 (1024 x 24) at its lower border: int32 keys, descending
])
def test_two_level_plan_typed_and_descending_exact_default_routing(gpu, oracle, n, kt, order, vb):
    _exact_case(gpu, oracle, 28 + kt + 2 * order, 0, vb, order=order, kt=kt, n=n, two_level=True)


def test_maximum_size_2pow30_minus_1_exact_vs_oracle(gpu, oracle):
    """The two-level plan's largest class (n > 2^29: 1024 x 24 = 24 576-key buckets) at the largest n the API accepts, EXACT against
    the oracle's sort (test_maximum_size_2pow30_minus_1 holds properties only)."""
    _exact_case(gpu, oracle, 30, 0, 0, n=(1 << 30) - 1, two_level=True)


def test_2pow28_ballot_ranking_takes_the_lsd_plan_exact(gpu, oracle):
    """rank_mode 0 (the ballot multi-split: what a part that fails gs_selftest_lds_atomic_order runs) at the headline
    size: the two-level plan, position chains and the pairs' bucket sort exist for LDS-atomic ranking only, so the sort must take the
    four LSD passes — and stay exact."""
    _exact_case(gpu, oracle, 28, 0, 0, rank_mode=0, two_level=False)


@pytest.mark.parametrize("andc", [0, 1, 2, 3, 4])
def test_2pow28_keys_exact_vs_oracle(gpu, oracle, andc):
    """configs[1] bit-exact at full size, at every entropy preset of the reference's sweep (GPUSortingD3D12/Tests.h:383-387):
    preset 1 on the library's default plan for uniform keys, presets 2..5 on the position-chain plan."""
    _exact_case(gpu, oracle, 28, andc, 0)


def test_2pow28_pairs_u32_index_exact_vs_oracle(gpu, oracle):
    """configs[2] bit-exact with value = index: keys, and the stable payload order (1024 x 16 fused tiles)."""
    _exact_case(gpu, oracle, 28, 0, 4)


@pytest.mark.parametrize("andc", [0, 1, 2, 3, 4])
def test_2pow28_pairs_u64_index_exact_vs_oracle(gpu, oracle, andc):
    """configs[4] bit-exact with value = index (u64) at ALL FIVE entropy presets of the reference's sweep
    (GPUSortingD3D12/Tests.h:383-387,406-410): the default routing sends presets 2..5 to the position-chain
    kernels (digit_binning_posv_kernel<8>), preset 1 to the plain form (512 x 32 tiles, late value fetch)."""
    _exact_case(gpu, oracle, 28, andc, 8)


@pytest.mark.parametrize("andc", [1, 2, 4])
def test_2pow28_pairs_u32_index_exact_presets_2_3_5(gpu, oracle, andc):
    """(u32, u32) pairs at entropy presets 2, 3 and 5, value = index (preset 1: above; preset 4: below)."""
    _exact_case(gpu, oracle, 28, andc, 4)


def test_2pow28_keys_exact_lsd_plan_only(gpu, oracle):
    """configs[1] with the two-level plan switched off (plan = 1): the reference's structure — GlobalHistogram, Scan, four LSD
    DigitBinningPasses — stays bit-exact at the headline size (the default plan's result is checked above)."""
    _exact_case(gpu, oracle, 28, 0, 0, plan=1)


def test_2pow26_pairs_ballot_ranking_exact_vs_oracle(gpu, oracle):
    """RANK 0 (64-lane ballot multi-split, the guaranteed path) at 2^26 with value = index."""
    _exact_case(gpu, oracle, 26, 0, 4, rank_mode=0)


def test_2pow27_pairs_descending_float_exact_vs_oracle(gpu, oracle):
    """The reference's middle size (2^27) through typed keys + descending order with value = index."""
    _exact_case(gpu, oracle, 27, 1, 4, order=1, kt=2)


def test_2pow28_pairs_u32_skewed_index_exact_vs_oracle(gpu, oracle):
    """configs[2] at entropy preset 4 with value = index: (u32, u32) pairs of skewed keys run the position-chain plan at its default
    threshold (round 3: values staged behind the keys in that plan's kernels) — keys and the stable payload order, bit-exact."""
    _exact_case(gpu, oracle, 28, 3, 4)


def test_2pow27_uint64_keys_exact(gpu):
    """64-bit keys at full size: 2^27 uniform uint64 keys, one histogram sweep + eight passes, against numpy's sort;
    then keys below 2^40 (three constant bytes: two passes dropped) by the order-aware Validate and a permutation checksum."""
    import torch
    n = 1 << 27
    g = torch.Generator(device="cuda")
    g.manual_seed(2027)
    dk = torch.randint(-(1 << 63), (1 << 63) - 1, (n,), dtype=torch.int64, device="cuda", generator=g)
    keys = dk.cpu().numpy().view(np.uint64).copy()
    s = gpu.OneSweep(n, key_type=gpu.KEY_UINT64)
    s.sort(dk)
    s.check()
    keys.sort()
    assert bool((dk == torch.from_numpy(keys.view(np.int64)).cuda()).all().item()), "sorted 64-bit keys differ from numpy's"
    del keys
    dk &= (1 << 40) - 1
    dk = dk[torch.randperm(n, device="cuda", generator=g)]
    before = (int(dk.sum().item()), int((dk % 1000003).sum().item()))
    s.sort(dk)
    s.check()
    r = s.check_state()
    assert sum(r["keys_per_pass"]) == 6 * n, r        # bytes 5..7 constant: one pair of passes dropped
    assert gpu.validate(dk, key_type=gpu.KEY_UINT64) == 0
    assert (int(dk.sum().item()), int((dk % 1000003).sum().item())) == before
    s.close()


@pytest.mark.parametrize("values", [2, 3])
def test_2pow28_few_valued_bytes_packed_counter_guard(gpu, values):
    """Every byte of the keys takes one of 2 (3) values with equal shares: the histogram kernel finds the digit groups uneven, the
    sort runs on position chains, and in every counting pass a workgroup's 16-bit next-digit counters would pass 65 535 several
    times over (2^28 / 512 workgroups = 524 288 keys each, half (a third) of them on a digit that is NOT the one left out) — the
    overflow guard of the packed table hands them to the global table in between.  Exact against torch.sort."""
    import torch
    n = 1 << 28
    g = torch.Generator(device="cuda"); g.manual_seed(2800 + values)
    keys = torch.zeros(n, dtype=torch.int64, device="cuda")
    for byte, vals in enumerate(((0x11, 0xfe, 0x80), (0x00, 0x7f, 0x3c), (0xa5, 0x5a, 0x01), (0x42, 0x41, 0xff))):
        pick = torch.randint(0, values, (n,), device="cuda", generator=g)
        lut = torch.tensor(vals[:values], dtype=torch.int64, device="cuda")
        keys |= lut[pick] << (8 * byte)
        del pick
    dk = keys.to(torch.int32)
    want = torch.sort(keys).values
    del keys
    s = gpu.OneSweep(n)
    s.sort(dk)
    s.check()
    assert bool(torch.equal(dk.to(torch.int64) & 0xFFFFFFFF, want))
    s.close()


@pytest.mark.parametrize("kind", ["sorted", "reverse_sorted", "block_clustered"])
def test_2pow28_presorted_and_clustered_inputs_exact(gpu, oracle, kind):
    """inputs the entropy presets do not cover, at the headline size, bit-exact.  Sorted and reverse-sorted keys (every
    wave of a tile holds ONE value of the upper bytes: the crowded-wave ranking of binning_body; every 4096-key chunk of the histogram
    sweep lies under one prefix) and block-clustered keys (2^20 consecutive positions share their top byte).  All three are
    near-uniform in their top 16 bits, so the device runs the two-level plan."""
    import torch
    n = 1 << 28
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.init_random(dk, 2828, 0)
    s = gpu.OneSweep(n)
    if kind == "block_clustered":
        idx = torch.arange(n, dtype=torch.int32, device="cuda")
        dk = ((dk & 0x00FFFFFF) | (((idx >> 20) * 37 & 0xFF) << 24)).contiguous()
        del idx
        keys = dk.cpu().numpy().view(np.uint32)
        want = torch.from_numpy(oracle.std_sort_parallel(keys, oracle.hardware_threads()).view(np.int32)).cuda()
    else:
        s.sort(dk)                        # (a sorted array to start from; its own exactness is test_2pow28_keys_exact_vs_oracle's business)
        assert gpu.validate(dk) == 0
        want = dk.clone()
        if kind == "reverse_sorted":
            dk = torch.flip(dk, dims=(0,)).contiguous()
    s.sort(dk)
    s.check()
    assert s.last_plan()["two_level"]
    assert bool((dk == want).all().item()), f"{kind}: result differs"
    s.close()


@pytest.mark.parametrize("vb", [0, 4, 8])
def test_3x2pow27_two_level_plan_against_the_lsd_passes(gpu, vb):
    """The two-level plan's third size class (2^28 < n <= 2^29: 12 288-element buckets, 1024-thread workgroups) on 3 x 2^27 elements:
    the same input sorted by the LSD passes (plan 1, exact against the oracle wherever the oracle reaches) and by the library's
    default must agree element for element — keys, and values = original index (stability)."""
    import torch
    n = 3 << 27
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.init_random(dk, 329, 0)
    out = []
    for plan in (1, 0):
        s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, value_bytes=vb, plan=plan)
        k = dk.clone()
        v = torch.arange(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda") if vb else None
        s.sort(k, v)
        s.check()
        assert s.last_plan()["two_level"] == (plan == 0)
        out.append((k, v))
        s.close()
    assert gpu.validate(out[1][0]) == 0
    assert bool((out[0][0] == out[1][0]).all().item()), "keys differ between the plans"
    if vb:
        assert bool((out[0][1] == out[1][1]).all().item()), "values differ between the plans"


@pytest.mark.parametrize("n,vb", [(3 << 24, 0), ((3 << 24) - 1, 0), (1 << 27, 0), ((1 << 27) + 1, 0), ((1 << 28) + 1, 0), ((1 << 25) + 1, 4), (1 << 27, 8),
                                  ((1 << 27) + 1, 4)])
def test_two_level_plan_at_its_thresholds_and_class_borders(gpu, n, vb):
    """The sizes at which the default routing changes: the plan's thresholds (keys 3 x 2^24, pairs 2^25 + 1) and the borders of the
    bucket-local sort's size classes (2^27 | 2^27 + 1, 2^28 | 2^28 + 1).  Default routing against the LSD passes (plan 1), element for
    element, values = original index."""
    import torch
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.init_random(dk, 4000 + (n & 0xFFFF), 0)   # (seed 0 would be the generator's degenerate state: every draw of a lane equal)
    out = []
    for plan in (1, 0):
        s = gpu.OneSweep(n, mode=gpu.MODE_PAIRS if vb else gpu.MODE_KEYS_ONLY, value_bytes=vb, plan=plan)
        k = dk.clone()
        v = torch.arange(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda") if vb else None
        s.sort(k, v)
        s.check()
        offered = n >= ((1 << 25) + 1 if vb else 3 << 24)
        assert s.last_plan()["two_level"] == (plan == 0 and offered), (n, vb, plan, s.last_plan())
        out.append((k, v))
        s.close()
    assert gpu.validate(out[1][0]) == 0
    assert bool((out[0][0] == out[1][0]).all().item())
    if vb:
        assert bool((out[0][1] == out[1][1]).all().item())


@pytest.mark.parametrize("log2n,pairs", [(28, False), (27, True)])
def test_sharded_pipeline_full_size_bucket_landed_bin_major_exact(gpu, oracle, log2n, pairs):
    """BASELINE configs[3]'s per-GPU share (2^28 keys) through gs_onesweep_sort_sharded on ONE rank with the exchange forced, default
    options: the bucket is offered the two-level plan, so it is landed in the alternate buffer and the local sort starts at the plan's
    second pass (gs_mgpu_last_layout) — exact against the oracle; pairs at 2^27 with value = index (stability through split, landing
    and the pre-grouped local sort)."""
    import os
    import torch
    import torch.distributed as dist
    from gpusorting_amd.sharded import ShardedOneSweep
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29537")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        n = 1 << log2n
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.init_random(dk, 2800 + log2n, 0)
        torch.cuda.synchronize()
        keys = dk.cpu().numpy().view(np.uint32)
        dv = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
        s = ShardedOneSweep(n, pairs=pairs, value_bytes=4, always_exchange=True, slack=1.0)
        bk, bv, nb = s.sort(dk, values=dv)
        s.check()
        assert nb == n and s.last_bin_major and s.engine.sorter.last_plan()["two_level"]
        if pairs:
            perm = torch.from_numpy(oracle.sort_permutation_parallel(keys, 0, 0).view(np.int32)).cuda()
            assert bool((bv[:n] == perm).all().item()), "payload order differs from the stable sort"
            assert bool((bk[:n] == dk[perm.to(torch.int64) & 0xFFFFFFFF]).all().item())
        else:
            ref = torch.from_numpy(oracle.std_sort_parallel(keys, oracle.hardware_threads()).view(np.int32)).cuda()
            assert bool((bk[:n] == ref).all().item()), "sorted keys differ from the oracle"
        s.close()
    finally:
        if created:
            dist.destroy_process_group()
