"""Fault injection (-m gpu): a build in which one tile never publishes its descriptor (the reference's
analogue: EmulatedDeadlocking.cu:36-37,339-345).  The sort must return — no hang — and report the timeout."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAULT_LIB = os.path.join(ROOT, "gpusorting_amd", "lib", "libgpusort_fault.so")


def test_withheld_descriptor_times_out_instead_of_hanging(gpu):
    if not os.path.exists(FAULT_LIB):
        pytest.fail("libgpusort_fault.so missing: run __graft_entry__.build()")
    code = textwrap.dedent("""
        import sys, time, torch
        sys.path.insert(0, %r)
        import gpusorting_amd as g
        n = 1 << 22                       # 256 tiles, 16 per chain: tile 5 of chain 3 stays silent in every pass
        k = torch.empty(n, dtype=torch.int32, device="cuda")
        g.init_random(k, 10, 0)
        s = g.OneSweep(n)
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
    """) % ROOT
    env = dict(os.environ, GPUSORT_LIB=FAULT_LIB)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RESULT status 4" in out.stdout, out.stdout + out.stderr[-500:]   # GS_ERR_TIMEOUT
    secs = float(out.stdout.split("seconds")[1].split()[0])
    assert secs < 20.0
    assert "SMALL True" in out.stdout
