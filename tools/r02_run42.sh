set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; tail -c 200 $O/bench.err; cut -c1-260 $O/bench.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 3 --warmup 1 --log2-keys 24 --dry-backend gloo > $O/bench_dry2.txt 2> $O/bench_dry2.err; tail -c 200 $O/bench_dry2.txt
timeout 600 python tools/size_sweep.py 30 > $O/size_sweep.txt 2>&1; sed -n 3,22p $O/size_sweep.txt
