set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
for p in 2 4; do python tools/shape_ab.py 4 $p auto 512x32 512x16 >> $O/shape_ab.txt 2>&1; done
for p in 2 4; do python tools/shape_ab.py 0 $p auto 512x16 256x32 >> $O/shape_ab.txt 2>&1; done
grep "^vb" $O/shape_ab.txt
