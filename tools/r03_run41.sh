cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_41; mkdir -p $O
export GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_tuning.so
for sh in 512x16 512x32 448x40; do
    echo "== GPUSORT_SHAPE=$sh (tuning build, general path)" >> $O/sweep.txt
    GPUSORT_SHAPE=$sh timeout 600 python tools/r03_midsweep.py 0 21 27 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
cat $O/sweep.txt
