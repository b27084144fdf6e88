cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_52; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-more"
rm -rf /tmp/p_sq1 /tmp/p_sq3
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -d /tmp/p_sq1 -- $CMD > $O/prof_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -d /tmp/p_sq3 -- python $R/tools/sort_loop.py 28 3 2 0 > $O/prof_sq3.log 2>&1
for d in p_sq1 p_sq3; do echo "### $d" >> $O/rocprof_sq.txt; python $R/tools/rocprof_summary.py $(find /tmp/$d -name "*.db" | head -1) >> $O/rocprof_sq.txt 2>&1; done
grep -E "global_histogram|digit_binning|hist_reduce" $O/rocprof_sq.txt | cut -c1-60,108-170
