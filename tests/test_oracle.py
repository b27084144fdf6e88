"""CPU tests: the oracle against the golden vectors and against itself (no GPU)."""
import os
import zlib

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "onesweep_golden.npz")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def _cases(golden):
    return [tuple(int(x) for x in row) for row in golden["cases"]]


def test_generator_matches_golden(oracle, golden):
    for ci, (n, seed, andc, kt, order, vb) in enumerate(_cases(golden)):
        keys = oracle.init_random(n, seed, andc)
        assert crc(keys) == int(golden[f"c{ci}_in_crc"]), f"case {ci}"
        np.testing.assert_array_equal(keys[:16], golden[f"c{ci}_in_head"])
        if f"c{ci}_in" in golden:
            np.testing.assert_array_equal(keys, golden[f"c{ci}_in"])


def test_generator_values_equal_keys(oracle):
    k, v = oracle.init_random(70000, 9, 1, value_bytes=4)
    np.testing.assert_array_equal(k, v)
    k, v = oracle.init_random(70000, 9, 1, value_bytes=8)
    np.testing.assert_array_equal(k.astype(np.uint64), v)


def test_sorts_match_golden(oracle, golden):
    for ci, (n, seed, andc, kt, order, vb) in enumerate(_cases(golden)):
        keys = oracle.init_random(n, seed, andc)
        vals = None if not vb else np.arange(n, dtype=np.uint32 if vb == 4 else np.uint64)
        for fn in (oracle.std_sort, oracle.onesweep_sort):
            r = fn(keys, kt, order, vals)
            sk, sv = (r, None) if vals is None else r
            assert crc(sk) == int(golden[f"c{ci}_out_crc"]), f"case {ci} {fn.__name__}"
            np.testing.assert_array_equal(sk[-16:], golden[f"c{ci}_out_tail"])
            if vb:
                assert crc(sv) == int(golden[f"c{ci}_vout_crc"]), f"case {ci} {fn.__name__} values"
            if f"c{ci}_out" in golden:
                np.testing.assert_array_equal(sk, golden[f"c{ci}_out"])


def test_entropy_presets_reduce_popcount(oracle):
    # Thearling-Smith: AND-ing k+1 uniform words leaves bit density 2^-(k+1)
    for andc, dens in enumerate((0.5, 0.25, 0.125, 0.0625, 0.03125)):
        k = oracle.init_random(1 << 16, 10, andc)
        ones = np.unpackbits(k.view(np.uint8)).mean()
        assert abs(ones - dens) < 0.01


@pytest.mark.parametrize("kt", [0, 1, 2])
def test_key_transform_is_order_preserving(oracle, kt):
    rng = np.random.default_rng(kt)
    raw = rng.integers(0, 1 << 32, size=4096, dtype=np.uint64).astype(np.uint32)
    if kt == 2:
        f = raw.view(np.float32)
        raw = raw[np.isfinite(f)]
    bits = np.array([oracle.lib.gso_key_to_bits(int(x), kt) for x in raw], dtype=np.uint32)
    back = np.array([oracle.lib.gso_bits_to_key(int(x), kt) for x in bits], dtype=np.uint32)
    np.testing.assert_array_equal(back, raw)
    native = raw if kt == 0 else raw.view(np.int32) if kt == 1 else raw.view(np.float32)
    order_native = np.argsort(native, kind="stable")
    order_bits = np.argsort(bits, kind="stable")
    # -0.0 < +0.0 in bit order but equal as floats: compare values, not permutations
    np.testing.assert_array_equal(native[order_native] == native[order_bits], True)


def test_structural_passes(oracle):
    keys = oracle.init_random(20000, 77, 0)
    hist = oracle.global_histogram(keys)
    assert hist.sum(axis=1).tolist() == [20000] * 4
    excl = oracle.scan(hist)
    np.testing.assert_array_equal(excl, np.cumsum(hist, axis=1) - hist)
    cur = keys
    for p in range(4):
        cur = oracle.digit_pass(cur, 8 * p)
        d = (cur >> (8 * p)) & 255
        assert np.all(d[:-1] <= d[1:])
    np.testing.assert_array_equal(cur, np.sort(keys))


def test_descending_is_reverse_of_stable_ascending(oracle):
    keys = oracle.init_random(5000, 3, 3)  # many duplicates
    vals = np.arange(5000, dtype=np.uint32)
    ak, av = oracle.onesweep_sort(keys, 0, 0, vals)
    dk, dv = oracle.onesweep_sort(keys, 0, 1, vals)
    np.testing.assert_array_equal(dk, ak[::-1])
    np.testing.assert_array_equal(dv, av[::-1])
    sk, sv = oracle.std_sort(keys, 0, 1, vals)
    np.testing.assert_array_equal(sk, dk)
    np.testing.assert_array_equal(sv, dv)


def test_validate_counts_inversions(oracle):
    a = np.array([1, 2, 2, 5, 4, 9, 8, 8], dtype=np.uint32)
    assert oracle.validate(a) == 2
    assert oracle.validate(np.sort(a)) == 0
    assert oracle.validate(np.sort(a)[::-1].copy(), order=1) == 0
    assert oracle.validate(np.sort(a), vals=np.sort(a)) == 0
    f = np.array([-1.5, -0.0, 0.0, 3.0], dtype=np.float32).view(np.uint32)
    assert oracle.validate(f, key_type=2) == 0
    assert oracle.validate(f) != 0  # as raw uint32 the negatives sort last


def test_config1_2pow16_host_path(oracle):
    """BASELINE.json configs[0]: 2^16 uint32 keys, host std::sort validation path (no GPU)."""
    keys = oracle.init_random(1 << 16, 10, 0)
    s = oracle.std_sort(keys)
    assert oracle.validate(s) == 0
    np.testing.assert_array_equal(s, np.sort(keys))
    np.testing.assert_array_equal(oracle.onesweep_sort(keys), s)
    np.testing.assert_array_equal(oracle.std_sort_parallel(keys, 4), s)


def test_parallel_sort(oracle):
    keys = oracle.init_random((1 << 18) + 11, 5, 0)
    for t in (1, 2, 3, 8):
        np.testing.assert_array_equal(oracle.std_sort_parallel(keys, t), np.sort(keys))


def test_msd_splitters(oracle):
    uni = np.full(256, 1000, dtype=np.uint64)
    assert oracle.msd_splitters(uni, 8).tolist() == [0, 32, 64, 96, 128, 160, 192, 224, 256]
    assert oracle.msd_splitters(uni, 1).tolist() == [0, 256]
    skew = np.zeros(256, dtype=np.uint64)
    skew[0] = 10**6
    fb = oracle.msd_splitters(skew, 4)
    assert fb[0] == 0 and fb[-1] == 256 and np.all(np.diff(fb.astype(np.int64)) >= 0)


# ---- the generator pinned against the REFERENCE's own code -----------------------------------
REF_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_init_random.npz")
REF_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_generator.so")


def test_generator_matches_reference_produced_vectors(oracle):
    """tests/golden/ref_init_random.npz was written by the reference's InitRandom kernels themselves
    (GPUSortingCUDA/UtilityKernels.cuh:53-117 compiled for the CPU, tests/golden/make_ref_golden.py)."""
    g = np.load(REF_GOLDEN)
    assert "UtilityKernels.cuh" in str(g["source"])
    for i, (n, seed, andc) in enumerate(g["cases"].tolist()):
        k = oracle.init_random(n, seed, andc)
        assert zlib.crc32(k.tobytes()) & 0xFFFFFFFF == int(g[f"crc_{i}"]), (n, seed, andc)
        np.testing.assert_array_equal(k[:64], g[f"head_{i}"])
        np.testing.assert_array_equal(k[-64:], g[f"tail_{i}"])
        kk, vv = oracle.init_random(n, seed, andc, 4)
        np.testing.assert_array_equal(kk, k)
        np.testing.assert_array_equal(vv, k)   # pairs overload: payload = key (:114-115)


@pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref is built only where /root/reference exists")
def test_generator_matches_reference_code_run_here(oracle):
    """oracle/_ref/libref_generator.so = the reference's kernels run one emulated thread at a time (oracle/Makefile)."""
    import ctypes as C
    ref = C.CDLL(REF_LIB)
    ref.ref_init_random.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    rng = np.random.default_rng(5)
    cases = [(1, 1, 0), (255, 9, 1), (65536, 10, 0), (65537, 11, 4), (300007, 12345, 2)]
    cases += [(int(rng.integers(1, 400000)), int(rng.integers(1, 1 << 31)), int(rng.integers(0, 5))) for _ in range(12)]
    for n, seed, andc in cases:
        k = np.empty(n, np.uint32)
        ref.ref_init_random(k.ctypes.data, None, andc, seed, n)
        np.testing.assert_array_equal(oracle.init_random(n, seed, andc), k, err_msg=f"n={n} seed={seed} and={andc}")


# ---- the sort pinned against the REFERENCE's own kernels ---------------------------------------
REF_SORT_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_onesweep.npz")
REF_SORT_LIB = os.path.join(os.path.dirname(REF_LIB), "libref_onesweep.so")


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def test_oracle_matches_reference_kernel_vectors(oracle):
    """tests/golden/ref_onesweep.npz was written by the reference's OneSweep kernels (GPUSortingCUDA/Sort/
    OneSweep.cu) executed on a CPU by the SIMT emulator of oracle/shim: global histogram, the key (and payload)
    buffer after every pass, the sorted result.  The oracle must reproduce every one of them."""
    g = np.load(REF_SORT_GOLDEN)
    assert "OneSweep.cu" in str(g["source"])
    for i, (n, seed, andc, pairs) in enumerate(g["cases"].tolist()):
        keys = oracle.init_random(n, seed, andc)
        assert oracle.validate(keys) == int(g[f"verr_{i}"]), (n, seed)   # the reference's Validate on the unsorted input
        assert _crc(oracle.global_histogram(keys)) == int(g[f"hist_{i}"]), (n, seed)
        cur_k, cur_v = keys.copy(), (np.arange(n, dtype=np.uint32) if pairs else None)
        for p in range(4):
            if pairs:
                cur_k, cur_v = oracle.digit_pass(cur_k, 8 * p, vals=cur_v)
                assert _crc(cur_v) == int(g[f"vcrc_{i}"][p]), (n, seed, p)
            else:
                cur_k = oracle.digit_pass(cur_k, 8 * p)
            assert _crc(cur_k) == int(g[f"kcrc_{i}"][p]), (n, seed, p)
        np.testing.assert_array_equal(cur_k[:32], g[f"head_{i}"])
        np.testing.assert_array_equal(cur_k[-32:], g[f"tail_{i}"])
        ref = oracle.std_sort(keys, 0, 0, np.arange(n, dtype=np.uint32) if pairs else None)
        if pairs:
            np.testing.assert_array_equal(ref[0], cur_k)
            np.testing.assert_array_equal(ref[1], cur_v)   # the reference's passes ARE the stable sort
        else:
            np.testing.assert_array_equal(ref, cur_k)
        np.testing.assert_array_equal(oracle.onesweep_sort(keys), cur_k)


@pytest.mark.skipif(not os.path.exists(REF_SORT_LIB), reason="oracle/_ref is built only where /root/reference exists")
def test_oracle_matches_reference_kernels_run_here(oracle):
    import ctypes as C
    ref = C.CDLL(REF_SORT_LIB)
    ref.ref_onesweep_sort_keys.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    ref.ref_onesweep_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    ref.ref_validate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    ref.ref_validate.restype = C.c_uint32
    rng = np.random.default_rng(11)
    for n in (2, 4096, 4097, 8192, 12289):   # the pass criterion, incl. its partition boundaries
        k = oracle.init_random(n, n, 0)
        assert ref.ref_validate(k.ctypes.data, None, n) == oracle.validate(k)
        v = k.copy()
        assert ref.ref_validate(k.ctypes.data, v.ctypes.data, n) == oracle.validate(k, 0, 0, v)
    for n in [1, 33, 7680, 15361] + [int(rng.integers(2, 30000)) for _ in range(4)]:
        andc = int(rng.integers(0, 5))
        keys = oracle.init_random(n, n + 3, andc)
        k = keys.copy()
        ref.ref_onesweep_sort_keys(k.ctypes.data, n, None, None)
        np.testing.assert_array_equal(k, oracle.std_sort(keys), err_msg=f"keys n={n}")
        k, v = keys.copy(), np.arange(n, dtype=np.uint32)
        ref.ref_onesweep_sort_pairs(k.ctypes.data, v.ctypes.data, n, None, None, None)
        rk, rv = oracle.std_sort(keys, 0, 0, np.arange(n, dtype=np.uint32))
        np.testing.assert_array_equal(k, rk, err_msg=f"pairs n={n}")
        np.testing.assert_array_equal(v, rv, err_msg=f"payload n={n}")


@pytest.mark.parametrize("kt", [0, 1, 2])
@pytest.mark.parametrize("order", [0, 1])
def test_sort_permutation_parallel_is_the_stable_sort(oracle, kt, order):
    """The full-size parity tests take the stable order from the parallel composite sort: it must equal
    std::stable_sort by key (descending: its exact reverse) — duplicates included (preset 4 keys)."""
    n = 200003
    keys = oracle.init_random(n, 11 + kt, 3)
    perm = oracle.sort_permutation_parallel(keys, kt, order, threads=8)
    rk, rv = oracle.std_sort(keys, kt, order, np.arange(n, dtype=np.uint32))
    np.testing.assert_array_equal(perm, rv)
    np.testing.assert_array_equal(keys[perm], rk)


def test_oracle_64bit_keys_against_numpy(oracle):
    """The 64-bit definitions the keys64 GPU tests check against: uint64 / int64 / float64 order, descending =
    reverse, stability, and eight stable byte passes == the sort."""
    rng = np.random.default_rng(3)
    n = 50021
    k = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    np.testing.assert_array_equal(oracle.std_sort64(k), np.sort(k))
    np.testing.assert_array_equal(oracle.std_sort64(k, 1).view(np.int64), np.sort(k.view(np.int64)))
    f = rng.standard_normal(n) * 1e100
    np.testing.assert_array_equal(oracle.std_sort64(f.view(np.uint64), 2).view(np.float64), np.sort(f))
    np.testing.assert_array_equal(oracle.std_sort64(f.view(np.uint64), 2, 1).view(np.float64), np.sort(f)[::-1])
    d = rng.integers(0, 5, size=n, dtype=np.uint64) << np.uint64(40)
    v = np.arange(n, dtype=np.uint32)
    rk, rv = oracle.std_sort64(d, 0, 0, v)
    perm = np.argsort(d, kind="stable")
    np.testing.assert_array_equal(rv, perm.astype(np.uint32))
    rk2, rv2 = oracle.std_sort64(d, 0, 1, v)
    np.testing.assert_array_equal(rv2, perm[::-1].astype(np.uint32))
    c, cv = k.copy(), v.copy()
    for p in range(8):
        c, cv = oracle.digit_pass64(c, 8 * p, 0, cv)
    ek, ev = oracle.std_sort64(k, 0, 0, v)
    np.testing.assert_array_equal(c, ek)
    np.testing.assert_array_equal(cv, ev)
