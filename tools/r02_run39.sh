set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final2; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $R/$O/prof_sq1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/$O/prof_sq2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $R/$O/prof_sq3 -- python $R/tools/sort_loop.py 28 3 2 0 > $R/$O/prof_sq3.log 2>&1
cd $R
for d in prof_sq1 prof_sq2 prof_sq3; do python tools/rocprof_summary.py $(find $O/$d -name "*_results.db") > $O/$d.txt 2>&1; done
rm -rf $O/prof_sq1 $O/prof_sq2 $O/prof_sq3
grep -E "digit_binning|global_hist" $O/prof_sq1.txt $O/prof_sq2.txt $O/prof_sq3.txt | cut -c1-60,150-260
