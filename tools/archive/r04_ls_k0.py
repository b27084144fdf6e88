"""Round 4: the first kernel of the local-sort plan alone under ablation bits (GPUSORT_LS_EXP, timing only: the passes behind it see garbage
tables, so only the first kernel's event slot is read and the device is reset by process exit)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ".")
    from gpusorting_amd import onesweep as osw
    n = 1 << 28
    k0 = torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device="cuda").to(torch.int32)
    s = osw.OneSweep(n); s.set_plan(True); s.set_profiling(True)
    best = 1e9
    for r in range(4):
        k = k0.clone(); torch.cuda.synchronize()
        s.sort(k); torch.cuda.synchronize()
        best = min(best, s.get_profile()["pass0"])
    print(f"LS_EXP={os.environ.get('GPUSORT_LS_EXP', '0'):>6}: first kernel {best:.4f} ms", flush=True)
else:
    for bits in (0, 256, 512, 1024, 2048, 4096, 256 | 512, 256 | 512 | 1024, 2048 | 256, 256 | 512 | 1024 | 4096):
        env = dict(os.environ, GPUSORT_LS_EXP=str(bits))
        subprocess.run([sys.executable, __file__, "child"], env=env, timeout=120)
