cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_25; mkdir -p $O
# extended fuzz: three more seeds x 120 cases x the three routings of tests/test_gpu_parity.py (position-chain plan allowed from 2^21 keys),
# then the same under the fault build (absent mid-route workgroups, a silent tile in every pass)
for seed in 777 4242 990011; do
GPUSORT_FUZZ_SEED=$seed GPUSORT_FUZZ_CASES=120 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz >> $O/fuzz.txt 2>&1
tail -2 $O/fuzz.txt
done
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_fault.so GPUSORT_FUZZ_SEED=31337 GPUSORT_FUZZ_CASES=100 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz >> $O/fuzz_fault.txt 2>&1
tail -2 $O/fuzz_fault.txt
