cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_54; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "low_entropy or skewed or 2pow28_keys_exact or pairs_u64" > $O/pytest_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_full.txt | tail -3
for pr in 1 2 4; do
  echo "== preset index $pr (entropy preset $((pr+1)))" >> $O/ab.txt
  timeout 900 python tools/ab.py gpusorting_amd/lib/libgpusort_prev.so gpusorting_amd/lib/libgpusort.so --vb 0,4,8 --rounds 2 --preset $pr 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
cat $O/ab.txt
