"""64-bit keys ("+ 64-bit keys (8 passes)"; the reference sorts 32-bit keys only, so the expected
result is the same DEFINITION on 8-byte keys: stable order by the radix-sortable bits, descending = exact reverse).
The HIP path sorts them with the 32-bit machinery on 8-byte elements: one GlobalHistogram sweep counts eight joint tables, one
Scan plans eight passes (round 3; GPUSORT_KEY64_SWEEPS=2: the two 4-pass rounds of round 2); every case is
bit-exact against the oracle (and the oracle's 64-bit functions against numpy in tests/test_oracle.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(a.view(np.int64 if a.dtype.itemsize == 8 else np.int32)).cuda()


def _keys(rng, n, kind):
    if kind == "uniform":
        return rng.integers(0, 2**64, size=n, dtype=np.uint64)
    if kind == "low32":          # high word constant: the whole second round is identity passes
        return rng.integers(0, 2**32, size=n, dtype=np.uint64)
    if kind == "high32":         # low word constant: the whole first round is identity passes
        return rng.integers(0, 2**32, size=n, dtype=np.uint64) << np.uint64(32)
    if kind == "dups":           # few distinct keys: stability shows in the payload
        return rng.integers(0, 37, size=n, dtype=np.uint64) * np.uint64(0x0101010101010101)
    if kind == "float":          # normals, zeros of both signs, infinities, denormals
        f = rng.standard_normal(n) * 10.0 ** rng.integers(-300, 300, size=n)
        f[::97] = 0.0
        f[1::97] = -0.0
        f[2::193] = np.inf
        f[3::193] = -np.inf
        return f.view(np.uint64).copy()
    raise ValueError(kind)


def _sort(gpu, keys, kt, order, vals, small_path=True, rank=None):
    s = gpu.OneSweep(keys.size, order, gpu.KEY_UINT64 + kt, gpu.MODE_KEYS_ONLY if vals is None else gpu.MODE_PAIRS,
                     0 if vals is None else vals.dtype.itemsize)
    s.set_small_path(small_path)
    if rank is not None:
        s.set_rank_mode(rank)
    dk = _dev(keys)
    dv = None if vals is None else _dev(vals)
    s.sort(dk, dv)
    s.check()
    r = s.check_state()
    assert (r["rows_not_inclusive"], r["rows_not_monotone"], r["chains_short_of_tickets"], r["hist_words_nonzero"]) == (0, 0, 0, 0), r
    if keys.size > 1:  # the order-aware Validate, 64-bit form: no inversion after the sort
        assert gpu.validate(dk, key_type=gpu.KEY_UINT64 + kt, order=order) == 0
    out = dk.cpu().numpy().view(np.uint64), (None if vals is None else dv.cpu().numpy().view(vals.dtype))
    s.close()
    return out


@pytest.mark.parametrize("kt", [0, 1, 2])
@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("vb", [0, 4, 8])
def test_keys64_sizes_types_orders(gpu, oracle, kt, order, vb):
    rng = np.random.default_rng(1000 + 100 * kt + 10 * order + vb)
    for n in (1, 2, 65, 1000, 8191, 8192, 8193, 20000, 70001, (1 << 20) + 3):
        keys = _keys(rng, n, "float" if kt == 2 else "uniform")
        vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        ref = oracle.std_sort64(keys, kt, order, vals)
        rk, rv = (ref, None) if vals is None else ref
        for small in ((True, False) if n <= 8192 else (True,)):
            ok, ov = _sort(gpu, keys, kt, order, vals, small_path=small)
            np.testing.assert_array_equal(ok, rk, err_msg=f"n={n} kt={kt} order={order} vb={vb} small={small}")
            if vb:
                np.testing.assert_array_equal(ov, rv, err_msg=f"values n={n} kt={kt} order={order} vb={vb} small={small}")


@pytest.mark.parametrize("kind", ["low32", "high32", "dups"])
@pytest.mark.parametrize("order", [0, 1])
def test_keys64_degenerate_words_and_stability(gpu, oracle, kind, order):
    """A constant word makes a whole round identity passes (dropped on the device, in pairs); few distinct keys put
    the stability of both rounds into the payload."""
    rng = np.random.default_rng(7)
    for n in (5000, 300000):
        keys = _keys(rng, n, kind)
        vals = np.arange(n, dtype=np.uint32)
        for rank in (0, 1):
            ok, ov = _sort(gpu, keys, 0, order, vals, rank=rank)
            rk, rv = oracle.std_sort64(keys, 0, order, vals)
            np.testing.assert_array_equal(ok, rk, err_msg=f"{kind} n={n} order={order} rank={rank}")
            np.testing.assert_array_equal(ov, rv, err_msg=f"{kind} values n={n} order={order} rank={rank}")


def test_keys64_each_of_the_eight_passes(gpu, oracle):
    """One stable DigitBinningPass per byte of the 64-bit key (structural entry point, pass 0..7)."""
    import torch
    rng = np.random.default_rng(11)
    n = 100003
    keys = _keys(rng, n, "uniform")
    vals = np.arange(n, dtype=np.uint32)
    s = gpu.OneSweep(n, key_type=gpu.KEY_UINT64, mode=gpu.MODE_PAIRS, value_bytes=4)
    dk, dv = _dev(keys), _dev(vals)
    for p in range(8):
        ok, ov = torch.zeros_like(dk), torch.zeros_like(dv)
        s.digit_pass(dk, ok, p, values_in=dv, values_out=ov)
        s.check()
        rk, rv = oracle.digit_pass64(keys, 8 * p, 0, vals)
        np.testing.assert_array_equal(ok.cpu().numpy().view(np.uint64), rk, err_msg=f"pass {p}")
        np.testing.assert_array_equal(ov.cpu().numpy().view(np.uint32), rv, err_msg=f"pass {p} values")
    s.close()


def test_keys64_2pow24_exact(gpu, oracle, routing):
    """A multi-thousand-tile case (2^24 + 12345 keys, value = index) and its throughput for the record."""
    import time
    import torch
    rng = np.random.default_rng(24)
    n = (1 << 24) + 12345
    keys = _keys(rng, n, "uniform")
    vals = np.arange(n, dtype=np.uint32)
    s = gpu.OneSweep(n, key_type=gpu.KEY_UINT64, mode=gpu.MODE_PAIRS, value_bytes=4)
    dk, dv = _dev(keys), _dev(vals)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.sort(dk, dv)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.check()
    rk, rv = oracle.std_sort64(keys, 0, 0, vals)
    np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint64), rk)
    np.testing.assert_array_equal(dv.cpu().numpy().view(np.uint32), rv)
    print(f"u64 keys + u32 values, n={n}: {dt * 1e3:.3f} ms (first call)")
    s.close()


def test_keys64_validate_counts_inversions(gpu, oracle):
    """gs_validate on 64-bit keys: the number of adjacent inversions of an unsorted array, per key type and order."""
    rng = np.random.default_rng(3)
    n = 100001
    for kt in (0, 1, 2):
        keys = _keys(rng, n, "float" if kt == 2 else "uniform")
        bits = np.array([oracle.lib.gso_key64_to_bits(int(k), kt) for k in keys[:2000]], dtype=np.uint64)
        part = _dev(np.ascontiguousarray(keys[:2000]))
        assert gpu.validate(part, key_type=gpu.KEY_UINT64 + kt, order=0) == int(np.count_nonzero(bits[:-1] > bits[1:]))
        assert gpu.validate(part, key_type=gpu.KEY_UINT64 + kt, order=1) == int(np.count_nonzero(bits[:-1] < bits[1:]))


def test_keys64_profile_slots_are_consistent(gpu, oracle):
    """ADVICE (round 2): the second round of a 64-bit sort re-recorded the events of the first, so slots went negative and the
    total covered half the sort.  Passes 4..7 are charged to slot 6 (pass 3): every slot >= 0, their sum == the total (the
    slots are consecutive event pairs), and the total is about twice a 32-bit sort's."""
    import torch
    n = 1 << 22
    rng = np.random.default_rng(5)
    dk = _dev(rng.integers(0, 2**64, size=n, dtype=np.uint64))
    s = gpu.OneSweep(n, gpu.ORDER_ASCENDING, gpu.KEY_UINT64)
    s.set_profiling(True)
    for _ in range(2):
        s.sort(dk)
        torch.cuda.synchronize()
    p = s.get_profile()
    parts = [p[k] for k in ("clear", "global_histogram", "scan", "pass0", "pass1", "pass2", "pass3")]
    assert all(x >= 0.0 for x in parts), p
    assert abs(sum(parts) - p["total"]) < 0.02 * p["total"] + 0.005, p
    assert p["pass3"] > 2.0 * p["pass0"], p   # passes 3..7
    s.close()


@pytest.mark.parametrize("sweeps", ["1", "2"])
def test_keys64_one_plan_for_eight_passes(gpu, oracle, monkeypatch, sweeps):
    """Round 3: ONE GlobalHistogram + Scan plans all eight passes of a 64-bit sort (the chains of pass 4 are the groups of byte 3);
    identity passes are dropped in pairs across the whole key.  GPUSORT_KEY64_SWEEPS=2 keeps the two-round form: same results."""
    monkeypatch.setenv("GPUSORT_KEY64_SWEEPS", sweeps)
    rng = np.random.default_rng(88)
    n = 300007
    cases = {
        "uniform": rng.integers(0, 2**64, size=n, dtype=np.uint64),
        "40 bits": rng.integers(0, 2**40, size=n, dtype=np.uint64),                      # bytes 5..7 constant: two of them dropped
        "bytes 2 and 5": (rng.integers(0, 256, size=n, dtype=np.uint64) << np.uint64(16)) | (rng.integers(0, 256, size=n, dtype=np.uint64) << np.uint64(40)),
        "constant": np.full(n, 0x0123456789abcdef, dtype=np.uint64),                     # every pass an identity
        "byte 3 skewed": (rng.integers(0, 2**32, size=n, dtype=np.uint64) << np.uint64(32)) | (rng.integers(0, 3, size=n, dtype=np.uint64) << np.uint64(28)) | rng.integers(0, 2**24, size=n, dtype=np.uint64),
    }
    for name, keys in cases.items():
        vals = np.arange(n, dtype=np.uint64)
        for order in (0, 1):
            s = gpu.OneSweep(n, order, gpu.KEY_UINT64, gpu.MODE_PAIRS, 8)
            dk, dv = _dev(keys), _dev(vals)
            s.sort(dk, dv)
            s.check()
            r = s.check_state()
            assert (r["rows_not_inclusive"], r["rows_not_monotone"], r["chains_short_of_tickets"], r["hist_words_nonzero"]) == (0, 0, 0, 0), (name, r)
            if sweeps == "1":  # one plan: keys_per_pass[q] sums passes q and q + 4
                ran = sum(r["keys_per_pass"]) // n
                expect = {"uniform": 8, "40 bits": 6, "bytes 2 and 5": 2, "constant": 2 if order else 0, "byte 3 skewed": 8}[name]
                assert ran == expect, (name, order, r)
            rk, rv = oracle.std_sort64(keys, 0, order, vals)
            np.testing.assert_array_equal(dk.cpu().numpy().view(np.uint64), rk, err_msg=f"{name} order={order}")
            np.testing.assert_array_equal(dv.cpu().numpy().view(np.uint64), rv, err_msg=f"{name} values order={order}")
            s.close()
