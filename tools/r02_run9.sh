set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
L=gpusorting_amd/lib
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_midpath.py tests/test_gpu_fault.py -m gpu -x -q > $O/pytest_sub.txt 2>&1
grep -E "passed|failed|rror" $O/pytest_sub.txt | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_full.txt 2>&1
grep -E "passed|failed|rror" $O/pytest_full.txt | tail -3
for p in 0 1 2 3 4; do timeout 300 python tools/ab.py $L/libgpusort.so $L/libgpusort_prev.so --rounds 2 --vb 0,8 --preset $p > $O/ab_preset$p.txt 2>&1; cat $O/ab_preset$p.txt; done
