cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_21; mkdir -p $O
echo "== default" > $O/mid.txt; timeout 300 python tools/r03_midsweep.py 0 >> $O/mid.txt 2>&1
echo "== GPUSORT_POS=2 GPUSORT_POS_MIN_LOG2=21" >> $O/mid.txt; GPUSORT_POS=2 GPUSORT_POS_MIN_LOG2=21 timeout 300 python tools/r03_midsweep.py 0 >> $O/mid.txt 2>&1
echo "== GPUSORT_MID_PATH=0 (general path below 2^22 too)" >> $O/mid.txt; GPUSORT_MID_PATH=0 timeout 300 python tools/r03_midsweep.py 0 >> $O/mid.txt 2>&1
echo "== pairs u32 default" >> $O/mid.txt; timeout 300 python tools/r03_midsweep.py 4 >> $O/mid.txt 2>&1
cat $O/mid.txt
