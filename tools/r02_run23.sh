set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O
timeout 300 python tools/shape_ab.py 0 0 auto 1024x32 512x32 > $O/shape_1024x32.txt 2>&1; cat $O/shape_1024x32.txt
timeout 300 python tools/shape_ab.py 0 2 auto 1024x32 >> $O/shape_1024x32.txt 2>&1; tail -2 $O/shape_1024x32.txt
