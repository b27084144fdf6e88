cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_18; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in min_old min; do
rm -rf /tmp/p_$lib
GPUSORT_LIB=$R/gpusorting_amd/lib/libgpusort_$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_$lib -- python $R/tools/sort_loop.py 28 6 > $O/$lib.log 2>&1
echo "### $lib" >> $O/stats.txt; python $R/tools/rocprof_summary.py $(find /tmp/p_$lib -name "*.db" | head -1) | cut -c1-60,110-160 >> $O/stats.txt 2>&1
done
cat $O/stats.txt
