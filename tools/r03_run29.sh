cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_29; mkdir -p $O
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp5120.so GPUSORT_MID_PATH=0 timeout 300 python tools/r03_hist_phases.py 22 25 2>&1 | grep -v amdgpu.ids > $O/hist_phases.txt
cat $O/hist_phases.txt
for vb in 4; do
  for lib in libgpusort_prev.so libgpusort.so; do
    echo "== vb=$vb $lib" >> $O/sweep.txt
    GPUSORT_LIB=$PWD/gpusorting_amd/lib/$lib timeout 600 python tools/r03_midsweep.py $vb 19 22 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
  done
done
cat $O/sweep.txt
