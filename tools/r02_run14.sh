set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
L=gpusorting_amd/lib
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -3
for p in 0 1 2 3 4; do timeout 300 python tools/ab.py $L/libgpusort.so $L/libgpusort_vr1.so --rounds 2 --vb 8 --preset $p > $O/ab_u64_p$p.txt 2>&1; cat $O/ab_u64_p$p.txt; done
