cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_20; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sharded_multirank.py tests/test_gpu_tools.py tests/test_capi.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
timeout 200 ./build/mgpu_main --gpus 1 --log2 28 --iters 5 > $O/mgpu_n1.json 2>&1; cat $O/mgpu_n1.json
timeout 200 ./build/mgpu_main --gpus 1 --log2 27 --iters 5 --pairs 8 --mode threads > $O/mgpu_n1_pairs8.json 2>&1; cat $O/mgpu_n1_pairs8.json
