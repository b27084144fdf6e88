"""The C++ side of the boundary under test: the reference's own programs, compiled against the C++ mirrors of its
dispatch surface (include/gpusort/*.hpp over the C-ABI), and the rocPRIM comparator, run as the driver would.

  build/gpusorting_main         GPUSortingCUDA/GPUSortingCUDA.cu:16-40 (TestAll* + BatchTiming*)
  build/gpusorting_d3d12_main   GPUSortingD3D12/Tests.h:6-186 (SuperTestOneSweep: 18 key x payload x order combinations)
  build/rocprim_compare         the AMD analogue of GPUSortingCUDA/Sort/CubDispatcher.cuh:105-404
"""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tools(gpu):
    """The binaries travel with the tree; (re)build whatever is missing or older than its sources."""
    subprocess.check_call(["make", "-C", ROOT, "-s", "tools"])
    return os.path.join(ROOT, "build")


def _run(cmd, timeout=600):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout + p.stderr


def test_reference_main_against_the_library(tools):
    """`GPUSortingCUDA.cu` main(): TestAllKeysOnly / TestAllPairs ([P, 2P] ladder + 2^26, 2^27, 2^28 = P + 4 sorts each)
    and the 2^28 batch timings, through OneSweepDispatcher.hpp.  Both halves must print 'All tests passed.'"""
    rc, out = _run([os.path.join(tools, "gpusorting_main"), "28", "20"])
    assert rc == 0, out[-2000:]
    passed = re.findall(r"(\d+)/(\d+) All tests passed\.", out)
    assert len(passed) == 2, out[-2000:]
    for a, b in passed:
        assert a == b and int(a) > 4096  # the whole ladder plus the three big sizes
    rates = [float(x) for x in re.findall(r"Estimated speed at \d+ 32-bit elements: ([0-9.E+]+) keys/sec", out)]
    assert len(rates) == 2 and rates[0] > 5e10 and rates[1] > 3e10, rates  # keys, pairs: far above any CPU or fallback rate
    print(f"gpusorting_main 28 20: keys {rates[0]:.3e} keys/s, pairs {rates[1]:.3e} pairs/s")


def test_d3d12_supertest_against_the_library(tools):
    """`SuperTestOneSweep` (Tests.h:6-186): {asc, desc} x {uint32, int32, float32 keys} x {uint32, int32, float32 payloads}."""
    rc, out = _run([os.path.join(tools, "gpusorting_d3d12_main"), "supertest"], timeout=900)
    assert rc == 0, out[-2000:]
    assert "18 / 18 ONESWEEP SUPER TEST PASSED!" in out, out[-2000:]


def test_rocprim_comparator_sorts_and_agrees_elementwise(tools):
    """The comparator (rocPRIM radix_sort_keys / _pairs, same generator and protocol) sorts, and on one 2^24 + 12345
    case its output equals this library's output element for element (keys; pairs with value = index)."""
    exe = os.path.join(tools, "rocprim_compare")
    rc, out = _run([exe, "check", "24"])
    assert rc == 0, out[-2000:]
    assert out.count("identical=yes") == 2, out
    rc, out = _run([exe, "26", "5"])
    assert rc == 0, out[-2000:]
    assert out.count("sorted=yes") == 2, out
    print(out)


@pytest.mark.parametrize("mode,pairs", [("fork", 0), ("threads", 4)])
def test_torch_free_multi_gpu_harness_on_one_gpu(tools, mode, pairs):
    """tools/mgpu_main.cpp (round-2 review, multi-GPU item (a)): one rank per GPU over the C-ABI, no Python —
    here with ONE rank whose exchange path is forced, in both process models (fork: one process per GPU with the communicator id
    handed through pipes; threads: the ncclCommInitAll model).  The JSON line must verify (sorted, sizes add up) and carry the
    phase times, the link figures and the local sort's roofline; with 8 GPUs the same binary runs `--gpus 8`."""
    import json
    rc, out = _run([os.path.join(tools, "mgpu_main"), "--gpus", "1", "--log2", "24", "--iters", "3", "--mode", mode, "--pairs", str(pairs)])
    assert rc == 0, out[-2000:]
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["verified"] is True and d["n_gpus"] == 1 and d["mode"] == mode and d["value_bytes"] == pairs, d
    assert d["value"] > 1.0 and d["phase_ms_max_over_ranks"]["local_sort"] > 0.0, d
    assert 0.0 < d["local_sort_rank0"]["roofline"]["frac"] < 1.0, d
    assert d["rank_exit_codes"] == [0], d


@pytest.mark.parametrize("pairs", [0, 4])
def test_world_8_rehearsal_on_one_gpu(tools, pairs):
    """the sharded pipeline end to end at the world size BASELINE.json configs[3] names, on a one-GPU box:
    `mgpu_main --ranks 8 --share-gpu` runs eight rank threads on device 0 over an in-process transport (a barrier and device-to-device
    copies per collective; RCCL refuses several ranks on one device) — the 8-way plan, 7 peers per rank, 8-entry count and
    displacement tables, the closing status gather, the local sorts.  The line must verify: every bucket sorted, sizes add up to
    8 x 2^22, bucket borders ascend across the ranks."""
    import json
    rc, out = _run([os.path.join(tools, "mgpu_main"), "--ranks", "8", "--share-gpu", "--log2", "22", "--iters", "2", "--pairs", str(pairs)])
    assert rc == 0, out[-2000:]
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert d["verified"] is True and d["n_gpus"] == 8 and d["value_bytes"] == pairs, d
    assert d["rank_exit_codes"] == [0] * 8 and d["links_used_per_rank"] == 7, d
    assert d["bytes_sent_off_rank"]["sum"] > 0.8 * 7 / 8 * 8 * (1 << 22) * (4 + pairs), d   # ~7/8 of every shard leaves its rank
