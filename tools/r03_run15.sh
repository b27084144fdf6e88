cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_15; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_midpath.py tests/test_gpu_fault.py tests/test_capi.py -m gpu -q -x > $O/pytest_mid.txt 2>&1
tail -25 $O/pytest_mid.txt
timeout 600 python tools/size_sweep.py 20 > $O/size_sweep.txt 2>&1; head -45 $O/size_sweep.txt
