cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_19; mkdir -p $O
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_ta0.so gpusorting_amd/lib/libgpusort_min_ta1.so --rounds 3 --vb 0 > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_ta0.so gpusorting_amd/lib/libgpusort_min_ta1.so --rounds 2 --vb 0 --preset 2 > $O/ab_p3.txt 2>&1; cat $O/ab_p3.txt
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min_ta1.so timeout 300 python tools/r03_pos_check.py 28 0 2>&1 | cut -c1-150 > $O/check.txt; cat $O/check.txt
