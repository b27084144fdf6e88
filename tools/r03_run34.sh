cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_34; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_midpath.py tests/test_gpu_fault.py tests/test_gpu_keys64.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
for vb in 0 4 8; do
  for lib in libgpusort_prev.so libgpusort.so; do
    echo "== vb=$vb $lib" >> $O/sweep.txt
    GPUSORT_LIB=$PWD/gpusorting_amd/lib/$lib timeout 600 python tools/r03_midsweep.py $vb 14 27 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
  done
done
cat $O/sweep.txt
