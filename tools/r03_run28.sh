cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_28; mkdir -p $O
# mid route v3: table read by the whole workgroup (16-byte sc1 loads), K2 spreads a bucket over all waves, 2^22 class on 256 tiles of 16 384
timeout 900 python -m pytest tests/test_gpu_midpath.py tests/test_gpu_fault.py -m gpu -q -x > $O/pytest_mid.txt 2>&1; tail -3 $O/pytest_mid.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not fuzz" > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
for vb in 0 4 8; do
  for lib in libgpusort_prev.so libgpusort.so; do
    echo "== vb=$vb $lib" >> $O/sweep.txt
    GPUSORT_LIB=$PWD/gpusorting_amd/lib/$lib timeout 600 python tools/r03_midsweep.py $vb 14 23 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
  done
done
cat $O/sweep.txt
