cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_30; mkdir -p $O
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp5120.so GPUSORT_MID_PATH=0 timeout 300 python tools/r03_hist_phases.py 23 24 2>&1 | grep -v amdgpu.ids > $O/hist_phases.txt
cat $O/hist_phases.txt
for lib in libgpusort_prev.so libgpusort_min.so; do
    echo "== vb=0 $lib" >> $O/sweep.txt
    GPUSORT_MID_PATH=0 GPUSORT_LIB=$PWD/gpusorting_amd/lib/$lib timeout 600 python tools/r03_midsweep.py 0 22 27 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
cat $O/sweep.txt
timeout 600 python tools/ab.py gpusorting_amd/lib/libgpusort_prev.so gpusorting_amd/lib/libgpusort_min.so --vb 0 --rounds 3 2>&1 | grep -v amdgpu.ids > $O/ab.txt
timeout 600 python tools/ab.py gpusorting_amd/lib/libgpusort_prev.so gpusorting_amd/lib/libgpusort_min.so --vb 0 --rounds 2 --preset 2 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
cat $O/ab.txt
