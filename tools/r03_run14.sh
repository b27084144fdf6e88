cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_14; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
