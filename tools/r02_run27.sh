set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "entropy or skew or heavy or preset or identity or exact" > $O/pytest_skew.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_skew.txt | tail -3
for L in libgpusort_skewold.so libgpusort.so; do GPUSORT_LIB=$PWD/gpusorting_amd/lib/$L timeout 300 python tools/entropy_breakdown.py 28 4 0,8 > $O/entropy_$L.txt 2>&1; cat $O/entropy_$L.txt; done
