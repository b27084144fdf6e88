set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_keys64.py -m gpu -x -q -s > $O/pytest_keys64.txt 2>&1
grep -E "passed|failed|rror|u64 keys" $O/pytest_keys64.txt | tail -5
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -3
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/prof_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/prof_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_write.log 2>&1
cd $R
for d in prof_stats prof_fetch prof_write; do python tools/rocprof_summary.py $(find $O/$d -name "*_results.db") > $O/$d.txt 2>&1; done
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench.txt 2> $O/bench.err
tail -c 300 $O/bench.err
