set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for p in 2 4; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_p$p -- python $R/tools/sort_loop.py 28 3 $p 0 > $R/$O/prof_p$p.log 2>&1
done
cd $R
for p in 2 4; do python tools/rocprof_summary.py $(find $O/prof_p$p -name "*_results.db") > $O/prof_p$p.txt 2>&1; head -9 $O/prof_p$p.txt; done
rm -rf $O/prof_p2 $O/prof_p4
