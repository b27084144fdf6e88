cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_42; mkdir -p $O
# extended fuzz on the final tree: three fresh seeds x 120 cases x the three routings, then 100 cases under the fault build
for seed in 1234567 24680 97531; do
GPUSORT_FUZZ_SEED=$seed GPUSORT_FUZZ_CASES=120 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz >> $O/fuzz.txt 2>&1
tail -1 $O/fuzz.txt
done
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_fault.so GPUSORT_FUZZ_SEED=8642 GPUSORT_FUZZ_CASES=100 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz > $O/fuzz_fault.txt 2>&1
tail -1 $O/fuzz_fault.txt
