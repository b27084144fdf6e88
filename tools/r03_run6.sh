cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_6; mkdir -p $O
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_pf0.so gpusorting_amd/lib/libgpusort_min_pf1.so --rounds 3 --vb 0 > $O/ab.txt 2>&1; cat $O/ab.txt
for p in 2 4; do timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_pf0.so gpusorting_amd/lib/libgpusort_min_pf1.so --rounds 2 --vb 0 --preset $p > $O/ab_p$p.txt 2>&1; cat $O/ab_p$p.txt; done
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min_pf1.so python - <<'PY'
import torch, gpusorting_amd as g
n=(1<<24)+12345
k=torch.empty(n,dtype=torch.int32,device="cuda")
for preset in range(5):
    g.init_random(k,7,preset); ref=torch.sort(k.view(torch.uint8).view(torch.int32).to(torch.int64)&0xffffffff).values
    s=g.OneSweep(n); s.sort(k); torch.cuda.synchronize()
    print("preset",preset,"exact",bool(((k.to(torch.int64)&0xffffffff)==ref).all()))
PY
