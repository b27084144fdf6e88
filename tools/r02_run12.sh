set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
L=gpusorting_amd/lib
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fault.py -m gpu -x -q > $O/pytest_sub.txt 2>&1
grep -E "passed|failed|rror" $O/pytest_sub.txt | tail -3
timeout 600 python tools/ab.py $L/libgpusort.so $L/libgpusort_vr1.so --rounds 3 --vb 8 > $O/ab_u64.txt 2>&1; cat $O/ab_u64.txt
timeout 600 python tools/ab.py $L/libgpusort.so $L/libgpusort_vr1.so --rounds 2 --vb 8 --preset 2 > $O/ab_u64_p3.txt 2>&1; cat $O/ab_u64_p3.txt
timeout 600 python tools/ab.py $L/libgpusort.so $L/libgpusort_vr1.so --rounds 2 --vb 8 --preset 4 > $O/ab_u64_p5.txt 2>&1; cat $O/ab_u64_p5.txt
timeout 600 python tools/ab.py $L/libgpusort.so $L/libgpusort_vr1.so --rounds 2 --vb 8 --log2 26 > $O/ab_u64_26.txt 2>&1; cat $O/ab_u64_26.txt
