cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_5; mkdir -p $O
export R03_MODES_ONLY=1
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp1024_ew.so R03_MODES=0,768,1792,0 timeout 300 python tools/r03_ablate.py 28 3 0 > $O/ew_modes.txt 2>&1; cat $O/ew_modes.txt
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min.so gpusorting_amd/lib/libgpusort_min_ew.so --rounds 3 --vb 0 > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min.so gpusorting_amd/lib/libgpusort_min_ew.so --rounds 2 --vb 0 --preset 2 > $O/ab_p3.txt 2>&1; cat $O/ab_p3.txt
