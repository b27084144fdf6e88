cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_37; mkdir -p $O
for pr in 1 2 3; do
  echo "== preset index $pr (entropy preset $((pr+1)))" >> $O/ab.txt
  timeout 900 python tools/ab.py gpusorting_amd/lib/libgpusort.so gpusorting_amd/lib/libgpusort_skew4.so --vb 0,4,8 --rounds 2 --preset $pr 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
cat $O/ab.txt
