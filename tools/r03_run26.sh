cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_25; mkdir -p $O
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_fault.so GPUSORT_FUZZ_SEED=31337 GPUSORT_FUZZ_CASES=100 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz > $O/fuzz_fault.txt 2>&1
tail -3 $O/fuzz_fault.txt
