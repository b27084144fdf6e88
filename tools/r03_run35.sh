cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_35; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_keys64.py tests/test_gpu_parity.py tests/test_gpu_sharded_multirank.py -m gpu -q -x -k "not fuzz" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 600 python tools/r03_keys64.py 27 4 2>&1 | grep -v amdgpu.ids > $O/keys64.txt
cat $O/keys64.txt
