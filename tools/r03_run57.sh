cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_57; mkdir -p $O
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time; tail -c 300 $O/bench.err; cut -c1-600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-more"
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- $CMD > $O/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -- $CMD > $O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -- $CMD > $O/prof_write.log 2>&1
for d in p_stats p_fetch p_write; do echo "### $d" >> $O/rocprof.txt; python $R/tools/rocprof_summary.py $(find /tmp/$d -name "*.db" | head -1) >> $O/rocprof.txt 2>&1; done
cut -c1-170 $O/rocprof.txt | head -70
