set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
L=gpusorting_amd/lib
GPUSORT_LIB=$PWD/$L/libgpusort_kr2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not fault" 2>&1 | tail -5 > $O/pytest_kr2.txt; cat $O/pytest_kr2.txt
timeout 400 python tools/ab.py $L/libgpusort.so $L/libgpusort_kr2.so --vb 0 --rounds 3 > $O/ab_kr2_p0.txt 2>&1; cat $O/ab_kr2_p0.txt
timeout 400 python tools/ab.py $L/libgpusort.so $L/libgpusort_kr2.so --vb 0 --rounds 2 --preset 3 > $O/ab_kr2_p3.txt 2>&1; cat $O/ab_kr2_p3.txt
