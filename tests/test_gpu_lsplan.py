"""GPU parity tests (-m gpu) of the LOCAL-SORT plan (gpusorting_amd/csrc/ls_kernels.hpp; gs_onesweep_set_plan): the fused
tile-local first kernel + gather pass + counting passes against the CPU oracle, bit-exact, and against the default
GlobalHistogram / Scan / 4-pass pipeline.  The plan is opt-in (default 0); plan 2 forces it at every size of the general
path, which is how the small and ragged sizes below reach it.

Reference behaviour: GPUSortingCUDA/Sort/OneSweep.cu:44-344 (four stable 8-bit passes, LSD); key types and descending
order: GPUSortingD3D12/Shaders/SortCommon.hlsl:134-154,594-597."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MAXK = (1 << 25) + 4096  # the handle owns the plan's tables only from 2^25 + 1 keys up


def to_dev(a):
    import torch
    return torch.from_numpy(a.view(np.int32)).cuda()


@pytest.fixture(scope="module")
def sorters(gpu):
    made = {}

    def get(kt, order):
        if (kt, order) not in made:
            s = gpu.OneSweep(MAXK, order, kt)
            s.set_small_path(False)
            s.set_mid_path(False)
            made[(kt, order)] = s
        return made[(kt, order)]
    yield get
    for s in made.values():
        s.close()


def _keys(oracle, n, seed, andc, kind):
    k = oracle.init_random(n, seed, andc, 0)
    if kind == "low16":
        k &= np.uint32(0xFFFF)
    elif kind == "high16":
        k &= np.uint32(0xFFFF0000)
    elif kind == "const":
        k[:] = 0xDEADBEEF
    elif kind == "sorted":
        k.sort()
    elif kind == "reversed":
        k[::-1].sort()
    elif kind == "blocks":      # long stretches of one low byte: a digit's runs fill whole source tiles
        k = (k & np.uint32(0xFFFFFF00)) | ((np.arange(n, dtype=np.uint32) // 40000) & np.uint32(0xFF))
    elif kind == "rare":        # low byte 0xFF about once per tile, everything else low byte 0: runs of one key
        k = np.where(k % 16384 == 0, k | np.uint32(0xFF), k & np.uint32(0xFFFFFF00)).astype(np.uint32)
    return k


@pytest.mark.parametrize("n", [1, 2, 63, 16383, 16384, 16385, 100003, (1 << 20) + 7, 3 * 16384 * 17])
@pytest.mark.parametrize("kt,order", [(0, 0), (0, 1), (1, 0), (2, 1)])
def test_local_sort_plan_small_and_ragged_sizes(gpu, oracle, sorters, n, kt, order):
    s = sorters(kt, order)
    k = _keys(oracle, n, n & 0xFFFF | 1, 0, "uniform")
    if kt == 2:
        k = np.where((k & 0x7F800000) == 0x7F800000, k & ~np.uint32(0x00800000), k).astype(np.uint32)  # (NaN patterns sort by bits in both)
    want = oracle.std_sort(k, kt, order)
    for plan in (2, 0):
        s.set_plan(plan)
        dk = to_dev(k.copy())
        s.sort(dk)
        s.check()
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), want, err_msg=f"plan {plan}")


@pytest.mark.parametrize("kind,andc", [("uniform", 1), ("uniform", 2), ("uniform", 4), ("low16", 0), ("high16", 0), ("const", 0),
                                       ("sorted", 0), ("reversed", 0), ("blocks", 0), ("rare", 0)])
def test_local_sort_plan_distributions(gpu, oracle, sorters, kind, andc):
    n = (1 << 22) + 12345
    k = _keys(oracle, n, 77, andc, kind)
    for order in (0, 1):
        s = sorters(0, order)
        want = oracle.std_sort(k, 0, order)
        s.set_plan(2)
        dk = to_dev(k.copy())
        s.sort(dk)
        s.check()
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), want)


def test_local_sort_plan_at_its_own_size_and_back_to_back(gpu, oracle, sorters):
    """Plan 1 (the size the plan is meant for), three sorts in a row on one handle — the tables and slices are reused — and a
    default-plan sort in between (it clears the slab regions the plan keeps its words in)."""
    import torch
    n = (1 << 25) + 4095
    s = sorters(0, 0)
    for rep, (andc, plan) in enumerate([(0, 1), (3, 1), (0, 0), (0, 1)]):
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.init_random(dk, 10 + rep, andc)
        k = dk.cpu().numpy().view(np.uint32)
        s.set_plan(plan)
        s.sort(dk)
        s.check()
        got = dk.cpu().numpy().view(np.uint32)
        assert gpu.validate(dk) == 0
        np.testing.assert_array_equal(got, np.sort(k))


def test_local_sort_plan_needs_its_tables(gpu):
    s = gpu.OneSweep(1 << 20)
    with pytest.raises(Exception):
        s.set_plan(1)      # max_keys <= 2^25: the handle has no tables for the plan
    s.set_plan(0)
    s.close()
    p = gpu.OneSweep(MAXK, mode=gpu.MODE_PAIRS, value_bytes=4)
    with pytest.raises(Exception):
        p.set_plan(1)      # pairs
    p.close()


def test_local_sort_plan_in_a_hip_graph(gpu, oracle, sorters):
    """No host round trip inside the plan either: captured once, replayed on new keys."""
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
    for seed in (6, 7):
        gpu.init_random(dk, seed, 1)
        torch.cuda.synchronize()
        k = dk.cpu().numpy().view(np.uint32)
        g.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint32), np.sort(k))
    s.check()
