cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_51; mkdir -p $O
L=gpusorting_amd/lib
timeout 900 python tools/ab.py $L/libgpusort_sleep0.so $L/libgpusort_sleep1.so $L/libgpusort_sleep2.so $L/libgpusort_sleep4.so --vb 0 --rounds 4 2>&1 | grep -v amdgpu.ids > $O/ab.txt
timeout 900 python tools/ab.py $L/libgpusort_sleep0.so $L/libgpusort_sleep1.so $L/libgpusort_sleep2.so $L/libgpusort_sleep4.so --vb 0 --rounds 2 --log2 24 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
cat $O/ab.txt
