cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
L=gpusorting_amd/lib
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "heavy_value" 2>&1 | grep -E "passed|failed" >> $O/stress_default.txt
  GPUSORT_LIB=$PWD/$L/libgpusort_vr1.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "heavy_value" 2>&1 | grep -E "passed|failed" >> $O/stress_vr1.txt
done
cat $O/stress_default.txt; echo ---; cat $O/stress_vr1.txt
