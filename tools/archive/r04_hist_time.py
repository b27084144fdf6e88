"""Round 4: GlobalHistogram kernel time (profile slot) of the default plan, 2^28 keys, entropy presets 1 and 3."""
import sys, os
import torch
sys.path.insert(0, ".")
import gpusorting_amd as g
n = 1 << 28
for preset in (0, 2):
    k = torch.empty(n, dtype=torch.int32, device="cuda")
    s = g.OneSweep(n); s.set_profiling(True)
    best = {}
    for r in range(5):
        g.init_random(k, 10 + r, preset); torch.cuda.synchronize()
        s.sort(k); torch.cuda.synchronize()
        p = s.get_profile()
        if r:
            for kk, v in p.items(): best[kk] = min(best.get(kk, 1e9), v)
    print(f"{os.path.basename(os.environ.get('GPUSORT_LIB', 'libgpusort.so')):36s} preset {preset + 1}: hist {best['global_histogram']:.4f} scan {best['scan']:.4f} pass0 {best['pass0']:.4f} pass1 {best['pass1']:.4f} total {best['total']:.4f}", flush=True)
    s.close()
