cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_31; mkdir -p $O
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp5120.so GPUSORT_MID_PATH=0 timeout 300 python tools/r03_hist_phases.py 23 24 2>&1 | grep -v amdgpu.ids > $O/hist_phases.txt
cat $O/hist_phases.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_midpath.py tests/test_gpu_fault.py tests/test_gpu_keys64.py -m gpu -q -x -k "not fuzz" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
for lib in libgpusort_prev.so libgpusort.so; do
    echo "== vb=0 $lib" >> $O/sweep.txt
    GPUSORT_LIB=$PWD/gpusorting_amd/lib/$lib timeout 600 python tools/r03_midsweep.py 0 22 27 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
for lib in libgpusort_prev.so libgpusort.so; do
    echo "== vb=4 $lib" >> $O/sweep.txt
    GPUSORT_LIB=$PWD/gpusorting_amd/lib/$lib timeout 600 python tools/r03_midsweep.py 4 22 25 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
cat $O/sweep.txt
timeout 600 python tools/ab.py gpusorting_amd/lib/libgpusort_prev.so gpusorting_amd/lib/libgpusort.so --vb 0,8 --rounds 3 2>&1 | grep -v amdgpu.ids > $O/ab.txt
cat $O/ab.txt
