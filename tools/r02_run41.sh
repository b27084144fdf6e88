cd $GRAFT_REPO_ROOT
O=gpurun_out/r02mid; mkdir -p $O
timeout 300 python tools/shape_by_size.py 0 21,22,23,24,25,26 512x32,512x16 > $O/shape_by_size.txt 2>&1
timeout 300 python tools/shape_by_size.py 4 22,23,24,25,26 1024x16,512x16,512x32 >> $O/shape_by_size.txt 2>&1
timeout 300 python tools/shape_by_size.py 8 21,22,23,24,25,26 512x32,512x16 >> $O/shape_by_size.txt 2>&1
cat $O/shape_by_size.txt
