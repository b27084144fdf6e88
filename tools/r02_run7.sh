set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
L=gpusorting_amd/lib
GPUSORT_LIB=$PWD/$L/libgpusort_slb.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_midpath.py -m gpu -x -q > $O/pytest_slb.txt 2>&1
grep -E "passed|failed|rror" $O/pytest_slb.txt | tail -3
timeout 900 python tools/ab.py $L/libgpusort.so $L/libgpusort_slb.so --rounds 3 --vb 0,4,8 > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 600 python tools/ab.py $L/libgpusort.so $L/libgpusort_slb.so --rounds 2 --vb 0 --preset 2 > $O/ab_preset3.txt 2>&1
cat $O/ab_preset3.txt
