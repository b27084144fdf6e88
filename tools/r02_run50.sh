cd $GRAFT_REPO_ROOT
O=gpurun_out/r02mid; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_midpath.py -m gpu -x -q 2>&1 | grep -E "passed|failed" > $O/pytest_mid2.txt; cat $O/pytest_mid2.txt
timeout 300 python - > $O/mid_times2.txt 2>&1 <<'P'
import torch, gpusorting_amd as g
for vb in (0, 4):
    for lg in (14, 16, 18, 20, 21, 22):
        n = 1 << lg
        if vb == 4 and lg == 22: continue
        k = torch.empty(n, dtype=torch.int32, device="cuda"); a = torch.empty_like(k)
        v = torch.empty(n, dtype=torch.int32, device="cuda") if vb else None; va = torch.empty_like(v) if vb else None
        s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
        ts = []
        for r in range(16):
            g.init_random(k, 7 + r, 0, v); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); s.sort(k, v, alt_keys=a, alt_values=va); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"vb={vb} 2^{lg} two-launch: {ts[len(ts)//2]:.1f} us (median of 16)")
        s.close()
P
cat $O/mid_times2.txt
