cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_27; mkdir -p $O
export TMPDIR=/tmp
# per-kernel timeline of mid-size sorts (kernel trace, csv) + histogram workgroup counts
for lg in 22 23 24 25; do
  for vb in 0 4; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_${lg}_${vb} -- python $GRAFT_REPO_ROOT/tools/r03_timeline.py run $lg $vb 3 > $O/run_${lg}_${vb}.log 2>&1)
    f=$(find $O/tr_${lg}_${vb} -name '*kernel_trace.csv' | head -1)
    python tools/r03_timeline.py parse $f >> $O/timeline.txt 2>&1
    rm -rf $O/tr_${lg}_${vb}
  done
done
cat $O/timeline.txt
for hb in 128 256 512 1024; do
  echo "== GPUSORT_HIST_BLOCKS=$hb" >> $O/histblocks.txt
  GPUSORT_HIST_BLOCKS=$hb timeout 300 python tools/r03_midsweep.py 0 2>&1 | grep -v amdgpu.ids >> $O/histblocks.txt
done
for hb in 256 512 768 1024; do
  echo "== 2^28 GPUSORT_HIST_BLOCKS=$hb" >> $O/histblocks.txt
  GPUSORT_HIST_BLOCKS=$hb timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort.so --vb 0 --rounds 2 2>&1 | grep -v amdgpu.ids >> $O/histblocks.txt
done
cat $O/histblocks.txt
