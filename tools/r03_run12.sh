cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_12; mkdir -p $O
GPUSORT_POS=0 timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_base.so gpusorting_amd/lib/libgpusort_min.so gpusorting_amd/lib/libgpusort_min_noor.so --rounds 2 --vb 0 > $O/ab_pos0.txt 2>&1; cat $O/ab_pos0.txt
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_base.so gpusorting_amd/lib/libgpusort_min.so --rounds 2 --vb 0 > $O/ab_pos1.txt 2>&1; cat $O/ab_pos1.txt
GPUSORT_POS=2 GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min.so timeout 600 python tools/r03_pos_check.py 28 0 2>&1 | cut -c1-150 > $O/forced.txt; cat $O/forced.txt
