set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
L=gpusorting_amd/lib
GPUSORT_LIB=$PWD/$L/libgpusort_jr.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "histogram or entropy or heavy or fuzz or degenerate or dropped or reference_kernel" > $O/pytest_jr.txt 2>&1
grep -E "passed|failed|rror" $O/pytest_jr.txt | tail -3
for p in 0 1 2 3 4; do timeout 300 python tools/ab.py $L/libgpusort.so $L/libgpusort_jr.so --rounds 2 --vb 0 --preset $p > $O/ab_p$p.txt 2>&1; cat $O/ab_p$p.txt; done
