cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_13; mkdir -p $O
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_base.so gpusorting_amd/lib/libgpusort_min.so --rounds 2 --vb 0 > $O/ab_pos1.txt 2>&1; cat $O/ab_pos1.txt
export GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min.so
for pos in 1 2; do echo "== GPUSORT_POS=$pos" >> $O/pos.txt
GPUSORT_POS=$pos timeout 600 python tools/r03_pos_check.py 28 0 2>&1 | cut -c1-150 >> $O/pos.txt; done
echo "== 2^26+777 POS=1" >> $O/pos.txt
timeout 600 python tools/r03_pos_check.py 26 777 2>&1 | cut -c1-150 >> $O/pos.txt
cat $O/pos.txt
