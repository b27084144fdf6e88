set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_1; mkdir -p $O
export GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp1024.so
timeout 300 python tools/r03_ablate.py 28 5 0 > $O/ablate_p1.txt 2>&1; cat $O/ablate_p1.txt
timeout 300 python tools/r03_ablate.py 28 3 2 > $O/ablate_p3.txt 2>&1; cat $O/ablate_p3.txt
unset GPUSORT_LIB
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort.so gpusorting_amd/lib/libgpusort_min.so gpusorting_amd/lib/libgpusort_exp1024.so --rounds 2 --vb 0 > $O/ab.txt 2>&1; cat $O/ab.txt
