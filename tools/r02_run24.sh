set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02y; mkdir -p $O
timeout 120 python - > $O/check.txt 2>&1 <<'P'
import torch, gpusorting_amd as g
n = (1 << 24) + 12345
for sh in ((768, 20), (640, 24)):
    k = torch.empty(n, dtype=torch.int32, device="cuda"); g.init_random(k, 5, 0)
    ref = torch.sort(k.to(torch.int64) & 0xffffffff).values
    s = g.OneSweep(n); s.set_shape(*sh); s.sort(k); torch.cuda.synchronize()
    print(sh, "sorted ok" if bool(((k.to(torch.int64) & 0xffffffff) == ref).all()) else "WRONG", s.check())
    s.close()
P
cat $O/check.txt
timeout 300 python tools/shape_ab.py 0 0 auto 768x20 640x24 > $O/shape_768.txt 2>&1; cat $O/shape_768.txt
timeout 300 python tools/shape_ab.py 0 2 auto 768x20 640x24 >> $O/shape_768.txt 2>&1; tail -3 $O/shape_768.txt
