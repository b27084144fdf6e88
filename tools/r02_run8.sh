set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
grep -E "passed|failed|rror" $O/pytest.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err ) 2> $O/bench_time.txt
tail -c 200 $O/bench.err; cat $O/bench_time.txt
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-more > $R/$O/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_mid -- python $R/tools/small_n.py 18 50 > $R/$O/prof_mid.log 2>&1
cd $R
for d in prof_stats prof_mid; do python tools/rocprof_summary.py $(find $O/$d -name "*_results.db") > $O/$d.txt 2>&1; done
rm -rf $O/prof_stats $O/prof_mid
./build/gpusorting_main 28 100 > $O/gpusorting_main.txt 2>&1; tail -12 $O/gpusorting_main.txt
./build/rocprim_compare 28 20 > $O/rocprim.txt 2>&1; cat $O/rocprim.txt
