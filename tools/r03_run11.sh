cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_11; mkdir -p $O
export GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min.so
for pos in 0 1 2; do
echo "== GPUSORT_POS=$pos" >> $O/pos.txt
GPUSORT_POS=$pos timeout 600 python tools/r03_pos_check.py 28 0 2>&1 | cut -c1-150 >> $O/pos.txt
done
cat $O/pos.txt
