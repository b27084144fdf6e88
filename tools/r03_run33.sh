cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_33; mkdir -p $O
for w in 0 25 27; do
    echo "== GPUSORT_WALK4_MAX_LOG2=$w libgpusort_min.so" >> $O/sweep.txt
    GPUSORT_WALK4_MAX_LOG2=$w GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min.so timeout 600 python tools/r03_midsweep.py 0 20 27 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
cat $O/sweep.txt
