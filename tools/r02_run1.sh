set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
./build/lds_microbench > $O/lds_microbench.txt 2>&1
L=gpusorting_amd/lib
timeout 600 python tools/ab.py $L/libgpusort_r01.so $L/libgpusort.so $L/libgpusort_ow0.so $L/libgpusort_exp1.so $L/libgpusort_exp256.so --rounds 3 --vb 0,4,8 > $O/ab.txt 2>&1
timeout 300 python tools/sweep.py 28 4 0 1 > $O/sweep_keys.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 300 python bench.py --steps 10 --warmup 2 > $O/bench.txt 2> $O/bench.err
tail -c 600 $O/bench.err
