cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_36; mkdir -p $O
export TMPDIR=/tmp
for lg in 16 18 20 21 22; do
  for vb in 0 8; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_${lg}_${vb} -- python $GRAFT_REPO_ROOT/tools/r03_timeline.py run $lg $vb 3 > $O/run_${lg}_${vb}.log 2>&1)
    f=$(find $O/tr_${lg}_${vb} -name '*kernel_trace.csv' | head -1)
    echo "-- 2^$lg vb=$vb" >> $O/timeline.txt
    python tools/r03_timeline.py parse $f >> $O/timeline.txt 2>&1
    rm -rf $O/tr_${lg}_${vb}
  done
done
cat $O/timeline.txt
