set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -3
