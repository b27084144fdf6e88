cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_55; mkdir -p $O
# fuzz on the last tree of the round (after the leave-one-out counting): two fresh seeds x 100 cases x three routings, then 60 cases under the fault build
for seed in 31415926 27182818; do
GPUSORT_FUZZ_SEED=$seed GPUSORT_FUZZ_CASES=100 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz >> $O/fuzz.txt 2>&1
tail -1 $O/fuzz.txt
done
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_fault.so GPUSORT_FUZZ_SEED=1618 GPUSORT_FUZZ_CASES=60 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k fuzz > $O/fuzz_fault.txt 2>&1
tail -1 $O/fuzz_fault.txt
