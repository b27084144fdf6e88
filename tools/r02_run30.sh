set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_keys64.py -m gpu -x -q -k "entropy or skew or heavy or preset or identity or exact or hist or constant or keys64" > $O/pytest_skew2.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_skew2.txt | tail -3
for L in libgpusort_skewold.so libgpusort.so; do GPUSORT_LIB=$PWD/gpusorting_amd/lib/$L timeout 300 python tools/entropy_breakdown.py 28 4 0 > $O/entropy2_$L.txt 2>&1; cat $O/entropy2_$L.txt; done
