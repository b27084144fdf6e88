set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 3 --warmup 1 --log2-keys 24 --dry-backend gloo > $O/bench_dry2.txt 2> $O/bench_dry2.err; tail -c 600 $O/bench_dry2.txt; tail -3 $O/bench_dry2.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 4 --steps 2 --warmup 1 --log2-keys 22 --pairs 8 --dry-backend gloo > $O/bench_dry4.txt 2> $O/bench_dry4.err; tail -c 600 $O/bench_dry4.txt; tail -3 $O/bench_dry4.err
