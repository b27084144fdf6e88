cd $GRAFT_REPO_ROOT
O=gpurun_out/r02mid; mkdir -p $O
timeout 300 python tools/shape_by_size.py 0 22,23,24,25 512x16,256x32,256x16,512x20 > $O/shape_by_size2.txt 2>&1
cat $O/shape_by_size2.txt
