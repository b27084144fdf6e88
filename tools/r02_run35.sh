set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final2; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/prof_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/prof_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/prof_fetch8 -- python $R/bench.py --pairs 8 --steps 3 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_fetch8.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/prof_write8 -- python $R/bench.py --pairs 8 --steps 3 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_write8.log 2>&1
cd $R
for d in prof_stats prof_fetch prof_write prof_fetch8 prof_write8; do python tools/rocprof_summary.py $(find $O/$d -name "*_results.db") > $O/$d.txt 2>&1; done
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_fetch8 $O/prof_write8
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; tail -c 200 $O/bench.err; cut -c1-300 $O/bench.txt
head -12 $O/prof_stats.txt
