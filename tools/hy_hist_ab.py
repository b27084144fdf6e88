import os, sys, subprocess
libs = sys.argv[1:]
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
import gpusorting_amd as g
n = 1 << 28
dk = torch.empty(n, dtype=torch.int32, device="cuda")
s = g.OneSweep(n, plan=2)
s.set_profiling(True)
hs, ts = [], []
for it in range(11):
    g.init_random(dk, 10 + it, 0)
    s.sort(dk)
    p = s.get_profile()
    if it: hs.append(p["global_histogram"]); ts.append(p["total"])
hs.sort(); ts.sort()
print(os.environ.get("GPUSORT_LIB", "product").split("/")[-1], "hist median %.4f min %.4f  total median %.4f" % (hs[len(hs)//2], hs[0], ts[len(ts)//2]))
'''
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, GPUSORT_LIB=os.path.join(os.getcwd(), "gpusorting_amd/lib", lib))
        print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
