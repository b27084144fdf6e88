cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_22; mkdir -p $O
for lib in min_base min; do
echo "== $lib" >> $O/mid.txt; GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_$lib.so timeout 300 python tools/r03_midsweep.py 0 >> $O/mid.txt 2>&1
done
cat $O/mid.txt
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_base.so gpusorting_amd/lib/libgpusort_min.so --rounds 3 --vb 0 > $O/ab.txt 2>&1; cat $O/ab.txt
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min.so timeout 300 python tools/r03_pos_check.py 28 0 2>&1 | cut -c1-150 > $O/check.txt; cat $O/check.txt
