cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_23; mkdir -p $O
for pos in 0 1; do echo "== pairs u64, GPUSORT_POS=$pos" >> $O/pairs8.txt
GPUSORT_POS=$pos timeout 600 python tools/r03_pos_check.py 28 0 8 2>&1 | cut -c1-150 >> $O/pairs8.txt; done
echo "== pairs u64 2^26+999 GPUSORT_POS=2 (forced)" >> $O/pairs8.txt
GPUSORT_POS=2 timeout 600 python tools/r03_pos_check.py 26 999 8 2>&1 | cut -c1-150 >> $O/pairs8.txt
cat $O/pairs8.txt
