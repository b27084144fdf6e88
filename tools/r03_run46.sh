cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_46; mkdir -p $O
export GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min.so
for v in 0 1 0 1; do
    echo "== GPUSORT_FIRST_PASS_BIG=$v" >> $O/sweep.txt
    GPUSORT_FIRST_PASS_BIG=$v timeout 600 python tools/r03_midsweep.py 0 22 26 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
cat $O/sweep.txt
