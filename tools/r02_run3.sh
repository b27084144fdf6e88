set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
L=gpusorting_amd/lib
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 900 python tools/ab.py $L/libgpusort.so $L/libgpusort_r01.so $L/libgpusort_h512.so $L/libgpusort_hu8.so $L/libgpusort_hu2.so $L/libgpusort_hnoprobe.so --rounds 3 --vb 0 > $O/ab_keys.txt 2>&1
cat $O/ab_keys.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 3 --warmup 1 --log2-keys 24 --dry-backend gloo > $O/bench_dry2.txt 2> $O/bench_dry2.err
tail -c 1500 $O/bench_dry2.txt; tail -5 $O/bench_dry2.err
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench.txt 2> $O/bench.err
tail -c 300 $O/bench.err
