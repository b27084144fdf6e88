cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_10; mkdir -p $O
for lib in min min_base; do
echo "== $lib" >> $O/pos.txt
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_$lib.so timeout 600 python tools/r03_pos_check.py 28 0 >> $O/pos.txt 2>&1
done
echo "== min 2^27+12345" >> $O/pos.txt
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min.so timeout 600 python tools/r03_pos_check.py 27 12345 >> $O/pos.txt 2>&1
cat $O/pos.txt
