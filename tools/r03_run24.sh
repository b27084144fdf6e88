cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_24; mkdir -p $O
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_rs0.so gpusorting_amd/lib/libgpusort_min_rs1.so --rounds 3 --vb 0 > $O/ab.txt 2>&1; cat $O/ab.txt
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min_rs1.so timeout 300 python tools/r03_pos_check.py 28 0 2>&1 | cut -c1-150 | head -2 > $O/check.txt; cat $O/check.txt
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min_rs1.so timeout 300 python tools/r03_pos_check.py 26 12345 2>&1 | cut -c1-150 | head -1 >> $O/check.txt; tail -1 $O/check.txt
