cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final3; mkdir -p $O
for seed in 777 4242; do GPUSORT_FUZZ_SEED=$seed GPUSORT_FUZZ_CASES=150 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k fuzz 2>&1 | grep -E "passed|failed|Error|case" | tail -3; done > $O/fuzz_long.txt 2>&1
cat $O/fuzz_long.txt
