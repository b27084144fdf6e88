"""Fault injection (-m gpu): builds in which one tile never publishes its descriptor, as if its workgroup had
stalled (the reference's analogue: EmulatedDeadlocking.cu:36-37,339-345).
  * libgpusort_fault.so            look-back fallback ON (the product's default): the successors recount the
                                   silent tile themselves, the sort is exact and reports GS_OK
                                   (reference: SweepCommon.hlsl:297-425 "look-back with fallback")
  * libgpusort_fault_nofallback.so fallback OFF: the sort must return — no hang — report GS_ERR_TIMEOUT and
                                   never write a key with a wrong prefix
"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "gpusorting_amd", "lib")


def _run(lib, code):
    path = os.path.join(LIBDIR, lib)
    if not os.path.exists(path):
        pytest.fail(f"{lib} missing: run __graft_entry__.build()")
    env = dict(os.environ, GPUSORT_LIB=path, GPUSORT_POS_MIN_LOG2="22")  # position-chain plan for skewed keys already at 2^22 keys
    out = subprocess.run([sys.executable, "-c", textwrap.dedent(code) % ROOT], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_withheld_descriptor_is_recounted_by_the_fallback(gpu):
    out = _run("libgpusort_fault.so", """
        import sys, time, torch
        sys.path.insert(0, %r)
        import gpusorting_amd as g
        # tile 5 of chain 3 stays silent in every pass; the third case is skewed (entropy preset 5): the position-chain
        # plan, where behind the first pass every chain's row 0 is seeded by its tile 0
        for n, pairs, preset in ((1 << 22, False, 0), ((1 << 22) + 12345, True, 0), ((1 << 23) + 777, False, 4)):
            k = torch.empty(n, dtype=torch.int32, device="cuda")
            g.init_random(k, 10, preset)
            v = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
            ref = torch.sort(k.to(torch.int64) & 0xffffffff, stable=True)
            s = g.OneSweep(n, mode=g.MODE_PAIRS if pairs else g.MODE_KEYS_ONLY, value_bytes=4 if pairs else 0)
            s.set_mid_path(False)          # 2^22 keys-only would take the two-launch route: no descriptors, nothing withheld
            t0 = time.time()
            s.sort(k, v)
            s.check()                      # raises on GS_ERR_TIMEOUT
            ok = bool(((k.to(torch.int64) & 0xffffffff) == ref.values).all().item())
            if pairs:
                ok = ok and bool((v.to(torch.int64) == ref.indices).all().item())
            print("RESULT", "exact" if ok else "WRONG", "seconds", round(time.time() - t0, 3))
    """)
    assert out.count("RESULT exact") == 3, out
    assert all(float(x.split()[0]) < 20.0 for x in out.split("seconds")[1:])


def test_withheld_descriptor_on_the_two_level_plan(gpu):
    """The two-level plan's DigitBinningPasses are the same kernels (binning_body): tile 5 of chain 3 stays silent in pass A (16 position
    chains) AND in pass B (256 chains: top-byte bucket 3 has eight tiles at 2^25 keys) — the look-back fallback recounts it, keys and
    (stable) pairs come out exact with GS_OK.  Without the fallback the sort returns GS_ERR_TIMEOUT instead of hanging."""
    code = """
        import sys, time, torch
        sys.path.insert(0, %r)
        import gpusorting_amd as g
        for n, pairs in (((1 << 25) + 999, False), ((1 << 25) + 5, True)):
            k = torch.empty(n, dtype=torch.int32, device="cuda")
            g.init_random(k, 10, 0)
            v = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
            ref = torch.sort(k.to(torch.int64) & 0xffffffff, stable=True)
            s = g.OneSweep(n, mode=g.MODE_PAIRS if pairs else g.MODE_KEYS_ONLY, value_bytes=4 if pairs else 0, plan=2)
            t0 = time.time()
            s.sort(k, v)
            try:
                s.check()
            except g.GpuSortError as e:
                print("RESULT status", e.status, "seconds", round(time.time() - t0, 3))
                continue
            ok = s.last_plan()["two_level"] and bool(((k.to(torch.int64) & 0xffffffff) == ref.values).all().item())
            if pairs:
                ok = ok and bool((v.to(torch.int64) == ref.indices).all().item())
            print("RESULT", "exact" if ok else "WRONG", "seconds", round(time.time() - t0, 3))
    """
    out = _run("libgpusort_fault.so", code)
    assert out.count("RESULT exact") == 2, out
    out = _run("libgpusort_fault_nofallback.so", code)
    assert out.count("RESULT status 4") == 2, out
    assert all(float(x.split()[0]) < 20.0 for x in out.split("seconds")[1:])


def test_mid_route_adopts_the_tiles_of_absent_workgroups(gpu):
    """Round-2 review, item 2 (reference: EmulatedDeadlocking.cu:36-37,339-345): in the fault build every fourth workgroup of the
    mid-size route's first kernel behaves as if it had never been dispatched.  The others adopt its tile — counts and
    scatter — so every size class of the route, both routes inside it (MSD + bucket sorts; the LSD passes for a skewed top
    byte), keys and pairs come out exact, with GS_OK."""
    out = _run("libgpusort_fault.so", """
        import sys, time, torch
        sys.path.insert(0, %r)
        import gpusorting_amd as g
        cases = [(20000, 0, False), (100003, 0, True), ((1 << 20) - 5, 0, False), ((1 << 21) + 5, 0, True), (1 << 22, 0, False),
                 (300000, 3, False), ((1 << 20) + 77, 3, True)]          # preset 4: the top byte is 0 for most keys -> LSD route
        for n, preset, pairs in cases:
            k = torch.empty(n, dtype=torch.int32, device="cuda")
            g.init_random(k, 21, preset)
            if preset:
                k &= 0x00FFFFFF                                          # one top-byte bucket holds everything
            v = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
            ref = torch.sort(k.to(torch.int64) & 0xffffffff, stable=True)
            s = g.OneSweep(n, mode=g.MODE_PAIRS if pairs else g.MODE_KEYS_ONLY, value_bytes=4 if pairs else 0)
            t0 = time.time()
            for rep in range(2):                                         # the handle's next epoch must work as well
                if rep: g.init_random(k, 21, preset); k &= (0x00FFFFFF if preset else -1)
                if rep and pairs: v.copy_(torch.arange(n, dtype=torch.int32, device="cuda"))
                s.sort(k, v)
                s.check()                                                # raises on GS_ERR_TIMEOUT
            ok = bool(((k.to(torch.int64) & 0xffffffff) == ref.values).all().item())
            if pairs:
                ok = ok and bool((v.to(torch.int64) == ref.indices).all().item())
            print("RESULT", "exact" if ok else "WRONG", n, "seconds", round(time.time() - t0, 3))
    """)
    assert out.count("RESULT exact") == 7, out


def test_withheld_descriptor_times_out_instead_of_hanging(gpu):
    out = _run("libgpusort_fault_nofallback.so", """
        import sys, time, torch
        sys.path.insert(0, %r)
        import gpusorting_amd as g
        n = 1 << 22                       # 512 tiles of 8192 keys, 32 per chain: tile 5 of chain 3 stays silent in every pass
        k = torch.empty(n, dtype=torch.int32, device="cuda")
        g.init_random(k, 10, 0)
        s = g.OneSweep(n)
        s.set_mid_path(False)             # the general pipeline (this size would take the two-launch route)
        t0 = time.time()
        s.sort(k)
        try:
            s.check()
            print("RESULT no-timeout")
        except g.GpuSortError as e:
            print("RESULT status", e.status, "seconds", round(time.time() - t0, 3))
        # the handle stays usable: a tiny sort goes through the single-tile kernel, no descriptors involved
        m = torch.randint(0, 1 << 30, (1000,), dtype=torch.int32, device="cuda")
        ref = torch.sort(m).values
        s2 = g.OneSweep(1000); s2.sort(m); torch.cuda.synchronize()
        print("SMALL", bool((m == ref).all().item()))
    """)
    assert "RESULT status 4" in out, out   # GS_ERR_TIMEOUT
    secs = float(out.split("seconds")[1].split()[0])
    assert secs < 20.0
    assert "SMALL True" in out


def test_mid_route_claimed_but_silent_tile_times_out_instead_of_hanging(gpu):
    """a tile of the mid-size route that was CLAIMED (so nobody can adopt it) but whose counts never appear.  The
    waiters' bounded spin must expire — GS_ERR_TIMEOUT, seconds, no hang — also while they keep trying to adopt.
    (The injected fault needs workgroup 2 of K1 to CLAIM its tile; once in a few hundred runs a neighbour adopts that tile before
    workgroup 2 is dispatched — then nothing is silent, the sort simply completes, and must be exact.  Seen once in round 6's full
    runs: each case is tried up to six times, every completed sort is validated, and a timeout must be seen.)"""
    out = _run("libgpusort_fault_nofallback.so", """
        import sys, time, torch
        sys.path.insert(0, %r)
        import gpusorting_amd as g
        for n, pairs in ((200000, False), ((1 << 20) + 3, True)):
            for attempt in range(6):
                k = torch.empty(n, dtype=torch.int32, device="cuda")
                g.init_random(k, 10 + attempt, 0)
                v = torch.arange(n, dtype=torch.int32, device="cuda") if pairs else None
                s = g.OneSweep(n, mode=g.MODE_PAIRS if pairs else g.MODE_KEYS_ONLY, value_bytes=4 if pairs else 0)
                t0 = time.time()
                s.sort(k, v)
                try:
                    s.check()
                    print("COMPLETED sorted", g.validate(k) == 0)
                except g.GpuSortError as e:
                    print("RESULT status", e.status, "seconds", round(time.time() - t0, 3))
                    break
    """)
    assert out.count("RESULT status 4") == 2, out   # GS_ERR_TIMEOUT, once per case
    assert all(float(x.split()[0]) < 20.0 for x in out.split("seconds")[1:])
    assert "COMPLETED sorted False" not in out, out
