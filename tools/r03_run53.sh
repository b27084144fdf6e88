cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_53; mkdir -p $O
L=gpusorting_amd/lib
for lib in libgpusort_minprev.so libgpusort_min.so; do
  echo "== $lib: tools/r03_pos_check.py 28 (exact vs torch.sort, presets 1..5)" >> $O/pos.txt
  GPUSORT_LIB=$PWD/$L/$lib timeout 600 python tools/r03_pos_check.py 28 2>&1 | grep -v amdgpu.ids >> $O/pos.txt
done
echo "== odd size, forced position chains (GPUSORT_POS=2), new build" >> $O/pos.txt
GPUSORT_POS=2 GPUSORT_LIB=$PWD/$L/libgpusort_min.so timeout 600 python tools/r03_pos_check.py 26 12345 2>&1 | grep -v amdgpu.ids >> $O/pos.txt
cat $O/pos.txt
