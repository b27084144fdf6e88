"""GPU parity tests (-m gpu) of the TWO-LEVEL plan (gpusorting_amd/csrc/hybrid_kernels.hpp; gs_onesweep_set_plan): one histogram
sweep over the keys' top 16 bits, a DigitBinningPass on the top byte, one on byte 2 inside the top-byte buckets (256 chains), and a
bucket-local sort of the low 16 bits in LDS — against the CPU oracle, bit-exact, and against the four LSD passes.  The device
decides per sort whether the plan applies (gs_onesweep_last_plan); keys it does not apply to must come out exact all the same,
through the LSD passes on position chains.  Plan 2 offers it at every size from 2^20 keys (position_chains_min_log2 = 20).

Reference behaviour: GPUSortingCUDA/Sort/OneSweep.cu:44-344 (histogram, scan, four stable 8-bit passes); key types and descending
order: GPUSortingD3D12/Shaders/SortCommon.hlsl:134-154,594-597."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MAXK = (1 << 25) + 4096


def to_dev(a):
    import torch
    return torch.from_numpy(a.view(np.int32)).cuda()


@pytest.fixture(scope="module")
def sorters(gpu):
    made = {}

    def get(kt, order):
        if (kt, order) not in made:
            made[(kt, order)] = gpu.OneSweep(MAXK, order, kt, small_path=0, mid_path=0, plan=2, position_chains_min_log2=20)
        return made[(kt, order)]
    yield get
    for s in made.values():
        s.close()


def _keys(oracle, n, seed, andc, kind):
    k = oracle.init_random(n, seed, andc, 0)
    if kind == "low16":        # one 16-bit prefix holds everything: the plan cannot apply
        k &= np.uint32(0xFFFF)
    elif kind == "high16":     # every bucket's keys are equal in their low 16 bits
        k &= np.uint32(0xFFFF0000)
    elif kind == "const":
        k[:] = 0xDEADBEEF
    elif kind == "sorted":
        k.sort()
    elif kind == "reversed":
        k[::-1].sort()
    elif kind == "blocks":      # long stretches of one top byte: whole tiles of pass A rank on one counter
        k = (k & np.uint32(0x00FFFFFF)) | (((np.arange(n, dtype=np.uint32) // 40000) & np.uint32(0xFF)) << np.uint32(24))
    elif kind == "top8":        # few top bytes in use, byte 2 uniform: 8 chains of pass B carry everything
        k = (k & np.uint32(0x07FFFFFF)) | ((k & np.uint32(7)) << np.uint32(29))
    elif kind == "bigbucket":   # uniform but for ONE 16-bit prefix that holds 1 % of the keys: above any workgroup's capacity
        k = np.where(k % 100 == 0, (k & np.uint32(0xFFFF)) | np.uint32(0x12340000), k).astype(np.uint32)
    return k


@pytest.mark.parametrize("n", [(1 << 20), (1 << 20) + 7, 3 * 16384 * 37, (1 << 22) + 12345, (1 << 24) - 1])
@pytest.mark.parametrize("kt,order", [(0, 0), (0, 1), (1, 0), (2, 1)])
def test_two_level_plan_sizes_types_orders(gpu, oracle, sorters, n, kt, order):
    s = sorters(kt, order)
    k = _keys(oracle, n, n & 0xFFFF | 1, 0, "uniform")
    if kt == 2:
        k = np.where((k & 0x7F800000) == 0x7F800000, k & ~np.uint32(0x00800000), k).astype(np.uint32)  # (NaN patterns sort by bits in both)
    want = oracle.std_sort(k, kt, order)
    for plan in (2, 1):
        s.set_plan(plan)
        dk = to_dev(k.copy())
        s.sort(dk)
        s.check()
        assert s.last_plan()["two_level"] == (plan == 2)
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), want, err_msg=f"plan {plan}")
        r = s.check_state()
        assert r["rows_not_inclusive"] == 0 and r["rows_not_monotone"] == 0 and r["chains_short_of_tickets"] == 0 and r["hist_words_nonzero"] == 0, r
        if plan == 2:
            assert r["keys_per_pass"][:2] == [n, n] and sum(r["keys_per_pass"]) == 2 * n, r   # pass A, pass B; LSD passes 2 and 3 did not run


@pytest.mark.parametrize("kind,andc,two_level", [("uniform", 0, True), ("uniform", 1, False), ("uniform", 2, False), ("uniform", 4, False),
                                                 ("low16", 0, False), ("high16", 0, True), ("const", 0, False), ("sorted", 0, True),
                                                 ("reversed", 0, True), ("blocks", 0, True), ("top8", 0, True), ("bigbucket", 0, False)])
def test_two_level_plan_distributions(gpu, oracle, sorters, kind, andc, two_level):
    """Whatever the keys look like the result is exact; WHICH plan ran is the device's decision, checked against what the
    distribution implies (a bucket above the local sort's capacity -> the LSD passes on position chains)."""
    n = (1 << 22) + 12345
    k = _keys(oracle, n, 77, andc, kind)
    for order in (0, 1):
        s = sorters(0, order)
        s.set_plan(2)
        want = oracle.std_sort(k, 0, order)
        dk = to_dev(k.copy())
        s.sort(dk)
        s.check()
        lp = s.last_plan()
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), want)
        assert lp["two_level"] == two_level, (kind, andc, lp)


def test_two_level_plan_back_to_back_with_other_plans(gpu, oracle, sorters):
    """Sorts in a row on one handle — uniform (two-level), skewed (falls back), LSD-only, uniform again: the slices, tables and slab
    regions are reused and every sort starts from a clean state."""
    import torch
    n = (1 << 25) + 4095
    s = sorters(0, 0)
    for rep, (andc, plan, two_level) in enumerate([(0, 2, True), (3, 2, False), (0, 1, False), (0, 2, True), (0, 0, False)]):
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.init_random(dk, 10 + rep, andc)
        k = dk.cpu().numpy().view(np.uint32)
        s.set_plan(plan)
        s.sort(dk)
        s.check()
        assert s.last_plan()["two_level"] == two_level, (rep, s.last_plan())   # (plan 0 offers it from 2^26 + 1 keys only)
        assert gpu.validate(dk) == 0
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), np.sort(k))


def test_two_level_plan_needs_its_tables(gpu, oracle):
    """A handle that the default routing can never send to the two-level plan (below its size, or created with plan 1) carries no tables
    for it (34 MB per handle saved); gs_onesweep_set_plan(2) allocates them on demand — and the next sort runs on the plan,
    exactly — except where the plan's fall-back does not exist either (max_keys <= 2^20)."""
    import torch
    s = gpu.OneSweep(1 << 20)
    with pytest.raises(Exception):
        s.set_plan(2)      # max_keys <= 2^20: neither the plan nor its fall-back (position chains) runs there
    s.set_plan(0)
    s.set_plan(1)
    s.close()
    for kwargs in ({"plan": 1}, {}):    # created with plan 1; created on default routing below the plan's size
        n = (1 << 22) + 321
        p = gpu.OneSweep(n, small_path=0, mid_path=0, position_chains_min_log2=20, **kwargs)
        keys = oracle.init_random(n, 77, 0)
        dk = torch.from_numpy(keys.view(np.int32)).cuda()
        p.sort(dk)
        p.check()
        assert not p.last_plan()["two_level"]
        p.set_plan(2)      # tables (and the larger histogram slices) allocated now
        dk = torch.from_numpy(keys.view(np.int32)).cuda()
        p.sort(dk)
        p.check()
        assert p.last_plan()["two_level"]
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), np.sort(keys))
        p.close()


# ---- pairs: the values go through both DigitBinningPasses with their keys and once more through the bucket-local sort ------------
@pytest.fixture(scope="module")
def pair_sorters(gpu):
    made = {}

    def get(vb, kt, order):
        if (vb, kt, order) not in made:
            made[(vb, kt, order)] = gpu.OneSweep(MAXK, order, kt, gpu.MODE_PAIRS, vb, small_path=0, mid_path=0, plan=2, position_chains_min_log2=20)
        return made[(vb, kt, order)]
    yield get
    for s in made.values():
        s.close()


@pytest.mark.parametrize("n", [(1 << 20) + 7, 3 * 16384 * 37, (1 << 23) + 4321])
@pytest.mark.parametrize("vb,kt,order", [(4, 0, 0), (4, 2, 1), (8, 0, 0), (8, 1, 1), (8, 0, 1)])
def test_two_level_plan_pairs_stable(gpu, oracle, pair_sorters, n, vb, kt, order):
    """value = original index, keys with many duplicates in their low bytes (masked to 20 significant bits spread over the word):
    keys AND the order of the values must be the stable sort's (descending: its exact reverse), on the two-level plan and on the LSD
    passes alike."""
    import torch
    s = pair_sorters(vb, kt, order)
    k = _keys(oracle, n, n & 0xFFFF | 1, 0, "uniform") & np.uint32(0xFFF00FFF if (n % 2 and kt != 2) else 0xFFFFFFFF)   # (odd n: 256-fold duplicates inside every bucket)
    if kt == 2:   # (NaN patterns sort by bits in both; folding them onto finite exponents of MASKED keys would double two prefixes past the bucket limit)
        k = np.where((k & 0x7F800000) == 0x7F800000, k & ~np.uint32(0x00800000), k).astype(np.uint32)
    v = np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
    wk, wv = oracle.std_sort(k, kt, order, v)
    for plan in (2, 1):
        s.set_plan(plan)
        dk = to_dev(k.copy())
        dv = torch.from_numpy(v.view(np.int32 if vb == 4 else np.int64).copy()).cuda()
        s.sort(dk, dv)
        s.check()
        assert s.last_plan()["two_level"] == (plan == 2)
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), wk, err_msg=f"plan {plan}")
        np.testing.assert_array_equal(dv.cpu().numpy().view(v.dtype), wv, err_msg=f"plan {plan}")


@pytest.mark.parametrize("vb", [4, 8])
@pytest.mark.parametrize("kind,andc,two_level", [("uniform", 2, False), ("high16", 0, True), ("sorted", 0, True), ("blocks", 0, True), ("bigbucket", 0, False)])
def test_two_level_plan_pairs_distributions(gpu, oracle, pair_sorters, vb, kind, andc, two_level):
    import torch
    n = (1 << 22) + 12345
    k = _keys(oracle, n, 78, andc, kind)
    v = np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
    for order in (0, 1):
        s = pair_sorters(vb, 0, order)
        s.set_plan(2)
        wk, wv = oracle.std_sort(k, 0, order, v)
        dk = to_dev(k.copy())
        dv = torch.from_numpy(v.view(np.int32 if vb == 4 else np.int64).copy()).cuda()
        s.sort(dk, dv)
        s.check()
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), wk)
        np.testing.assert_array_equal(dv.cpu().numpy().view(v.dtype), wv)
        assert s.last_plan()["two_level"] == two_level, (kind, s.last_plan())


def test_two_level_plan_in_a_hip_graph(gpu, oracle, sorters):
    """No host round trip inside the plan: captured once, replayed on new keys — uniform keys run the two-level plan, skewed keys the
    LSD passes, from the SAME captured launches."""
    import torch
    n = (1 << 21) + 999
    s = sorters(0, 0)
    s.set_plan(2)
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    alt = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.init_random(dk, 5, 0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s.sort(dk, alt_keys=alt)  # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s.sort(dk, alt_keys=alt)
    for seed, andc, two_level in ((6, 0, True), (7, 3, False), (8, 0, True)):
        gpu.init_random(dk, seed, andc)
        torch.cuda.synchronize()
        k = dk.cpu().numpy().view(np.uint32)
        g.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), np.sort(k))
        assert s.last_plan()["two_level"] == two_level
    s.check()


def test_two_level_plan_default_threshold_and_2pow27(gpu, oracle):
    """Plan 0 (the default): offered above 2^26 keys — 2^27 uniform keys run it, exact against the oracle's parallel sort."""
    import torch
    n = 1 << 27
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.init_random(dk, 27, 0)
    keys = dk.cpu().numpy().view(np.uint32)
    s = gpu.OneSweep(n)
    s.sort(dk)
    s.check()
    assert s.last_plan()["two_level"]
    ref = oracle.std_sort_parallel(keys, oracle.hardware_threads())
    assert bool((dk == torch.from_numpy(ref.view(np.int32)).cuda()).all().item())
    s.close()


@pytest.mark.parametrize("vb", [4, 8])
def test_two_level_plan_pairs_in_a_hip_graph(gpu, oracle, pair_sorters, vb):
    """Pairs: captured once, replayed on uniform keys (two-level plan) and on skewed keys (LSD passes on position chains) — the same
    captured launches, value = index exact both times."""
    import torch
    n = (1 << 21) + 4097
    s = pair_sorters(vb, 0, 0)
    s.set_plan(2)
    vdt = torch.int32 if vb == 4 else torch.int64
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    dv = torch.arange(n, dtype=vdt, device="cuda")
    alt, valt = torch.empty_like(dk), torch.empty_like(dv)
    gpu.init_random(dk, 5, 0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        s.sort(dk, dv, alt_keys=alt, alt_values=valt)  # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s.sort(dk, dv, alt_keys=alt, alt_values=valt)
    for seed, andc, two_level in ((6, 0, True), (7, 3, False), (8, 0, True)):
        gpu.init_random(dk, seed, andc)
        dv.copy_(torch.arange(n, dtype=vdt, device="cuda"))
        torch.cuda.synchronize()
        k = dk.cpu().numpy().view(np.uint32)
        perm = np.argsort(k, kind="stable")
        g.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), k[perm])
        np.testing.assert_array_equal(dv.cpu().numpy().astype(np.int64), perm)
        assert s.last_plan()["two_level"] == two_level
    s.check()


@pytest.mark.parametrize("heavy,two_level", [(3071, True), (3072, True), (3073, False), (40000, False)])
def test_two_level_plan_bucket_capacity_boundary(gpu, heavy, two_level):
    """The device's decision is exact: a 16-bit prefix holding exactly what the bucket-local sort's workgroup holds (3072 keys in the size
    class up to 2^27 keys) still runs the two-level plan; one key more and the same launches run the LSD passes.  Both results exact."""
    import torch
    n = (1 << 22) + 999
    g = torch.Generator(device="cuda"); g.manual_seed(heavy)
    k = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int64, device="cuda", generator=g) & 0xFFFFFFFF
    k = torch.where((k >> 16) == 0x1234, k ^ 0x00010000, k)                # nobody has prefix 0x1234 ...
    pos = torch.randperm(n, device="cuda", generator=g)[:heavy]
    k[pos] = (k[pos] & 0xFFFF) | (0x1234 << 16)                              # ... but exactly `heavy` keys
    want = torch.sort(k).values
    for order in (0, 1):
        s = gpu.OneSweep(n, order, small_path=0, mid_path=0, plan=2, position_chains_min_log2=20)
        dk = k.to(torch.int32)
        s.sort(dk)
        s.check()
        lp = s.last_plan()
        got = dk.to(torch.int64) & 0xFFFFFFFF
        assert bool(torch.equal(got, want if order == 0 else torch.flip(want, dims=(0,)))), (heavy, order)
        assert lp["two_level"] == two_level and (not two_level or lp["largest_bucket"] == max(heavy, lp["largest_bucket"])), (heavy, lp)
        s.close()


def test_two_level_plan_fuzz(gpu):
    """A seeded sweep of sizes and key shapes through the forced two-level plan — masks that empty most prefixes or most low bits, constant
    top bytes, mixtures of a narrow and a wide range, runs of equal keys — keys-only and (u32, u32) pairs with value = index, both orders,
    against torch's stable sort.  Whatever the device decides (two-level plan or its fall-back), the result must be exact."""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(20260926)
    rng = np.random.default_rng(505)
    ran = {True: 0, False: 0}
    for it in range(40):
        n = int(rng.integers(1 << 20, 1 << 23)) | 1
        k = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int64, device="cuda", generator=g) & 0xFFFFFFFF
        shape = it % 8
        if shape == 1:
            k &= int(rng.integers(0, 1 << 32)) | 0xFFFF0000                  # random low-bit mask: duplicates inside the buckets
        elif shape == 2:
            k &= 0x0FFFFFFF | (int(rng.integers(0, 16)) << 28)               # few top nibbles
        elif shape == 3:
            k = (k & 0x00FFFFFF) | (int(rng.integers(0, 256)) << 24)         # constant top byte: 256 prefixes hold everything
        elif shape == 4:
            narrow = torch.rand(n, device="cuda", generator=g) < 0.3
            k = torch.where(narrow, (k & 0xFFFFF) | 0x7A500000, k)            # 30 % of the keys inside sixteen prefixes
        elif shape == 5:
            k = (k >> 7) << 7                                                 # runs of up to 128 equal keys after sorting
        elif shape == 6:
            k = torch.sort(k).values                                          # presorted
        elif shape == 7:
            k &= 0xFFFF00FF                                                   # byte 1 constant: an identity pass inside every bucket
        pairs = it % 3 == 0
        order = (it // 3) % 2
        s = gpu.OneSweep(n, order, 0, gpu.MODE_PAIRS if pairs else gpu.MODE_KEYS_ONLY, 4 if pairs else 0, small_path=0, mid_path=0, plan=2,
                         position_chains_min_log2=20)
        dk = k.to(torch.int32)
        dv = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
        ref = torch.sort(k, stable=True)
        s.sort(dk, dv)
        s.check()
        ran[s.last_plan()["two_level"]] += 1
        wk, wi = (ref.values, ref.indices) if order == 0 else (torch.flip(ref.values, dims=(0,)), torch.flip(ref.indices, dims=(0,)))
        assert bool(torch.equal(dk.to(torch.int64) & 0xFFFFFFFF, wk)), (it, n, shape, pairs, order)
        if pairs:
            assert bool(torch.equal(dv.to(torch.int64), wi)), (it, n, shape, pairs, order, "values")
        s.close()
    assert ran[True] >= 10 and ran[False] >= 5, ran   # both outcomes were exercised
