cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_48; mkdir -p $O
for v in 0 1 0 1; do
    echo "== vb=4 GPUSORT_FIRST_PASS_BIG_VB4=$v" >> $O/sweep.txt
    GPUSORT_FIRST_PASS_BIG_VB4=$v timeout 600 python tools/r03_midsweep.py 4 22 24 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
cat $O/sweep.txt
