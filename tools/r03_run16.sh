cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_16; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt
timeout 600 python tools/size_sweep.py 20 > $O/size_sweep.txt 2>&1; head -24 $O/size_sweep.txt
