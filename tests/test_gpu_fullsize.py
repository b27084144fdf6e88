"""GPU tests at BASELINE.json's full sizes: exact parity at 2^26 against the CPU
oracle, size-independent properties at 2^28 (sortedness, permutation-invariant
checksums, digit histograms preserved)."""
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
