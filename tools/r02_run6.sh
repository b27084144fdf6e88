set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_midpath.py -m gpu -x -q -s > $O/pytest_mid.txt 2>&1
grep -E "passed|failed|rror|us per sort" $O/pytest_mid.txt | tail -14
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -3
timeout 300 python tools/size_sweep.py 10 > $O/size_sweep.txt 2>&1
head -45 $O/size_sweep.txt
