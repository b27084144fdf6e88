set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
L=gpusorting_amd/lib
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.txt 2>&1
tail -3 $O/pytest_parity.txt
timeout 900 python tools/ab.py $L/libgpusort.so $L/libgpusort_early0.so $L/libgpusort_early4.so $L/libgpusort_early16.so $L/libgpusort_walk4.so $L/libgpusort_histrep0.so $L/libgpusort_exp256.so $L/libgpusort_exp1.so --rounds 3 --vb 0 > $O/ab_keys.txt 2>&1
cat $O/ab_keys.txt
timeout 600 python tools/ab.py $L/libgpusort.so $L/libgpusort_early0.so --rounds 2 --vb 4,8 > $O/ab_pairs.txt 2>&1
cat $O/ab_pairs.txt
for lib in libgpusort.so libgpusort_histrep0.so; do GPUSORT_LIB=$PWD/$L/$lib timeout 300 python tools/entropy_breakdown.py 28 3 0 > $O/entropy_$lib.txt 2>&1; done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
