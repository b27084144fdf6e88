cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final3; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
for lg in 21 22; do timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_mid$lg -- python $R/tools/small_n.py $lg 50 > $R/$O/prof_mid$lg.log 2>&1; done
cd $R
for lg in 21 22; do python tools/rocprof_summary.py $(find $O/prof_mid$lg -name "*_results.db") > $O/prof_mid$lg.txt 2>&1; head -8 $O/prof_mid$lg.txt; done
rm -rf $O/prof_mid21 $O/prof_mid22
