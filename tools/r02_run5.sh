set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
L=gpusorting_amd/lib
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fault.py tests/test_gpu_keys64.py tests/test_gpu_sharded_multirank.py -m gpu -x -q > $O/pytest_sub.txt 2>&1
grep -E "passed|failed|rror" $O/pytest_sub.txt | tail -3
timeout 900 python tools/ab.py $L/libgpusort.so $L/libgpusort_ticket.so $L/libgpusort_hnoprobe.so $L/libgpusort_r01.so --rounds 3 --vb 0 > $O/ab_keys.txt 2>&1
cat $O/ab_keys.txt
timeout 600 python tools/ab.py $L/libgpusort.so $L/libgpusort_ticket.so --rounds 2 --vb 4,8 > $O/ab_pairs.txt 2>&1
cat $O/ab_pairs.txt
timeout 600 python tools/ab.py $L/libgpusort.so $L/libgpusort_ticket.so --rounds 2 --vb 0 --preset 2 > $O/ab_keys_preset3.txt 2>&1
cat $O/ab_keys_preset3.txt
timeout 300 python tools/size_sweep.py 10 > $O/size_sweep.txt 2>&1
tail -40 $O/size_sweep.txt
