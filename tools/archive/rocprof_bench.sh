#!/bin/bash
# rocprofv3 passes of the bench command for profiles/: kernel stats, then FETCH_SIZE and WRITE_SIZE in runs of their own
# (counters never together with a trace domain other than --kernel-trace).  usage (GPU box): tools/rocprof_bench.sh OUT_PREFIX [bench args]
set -u
out=${1:-gpurun_out/rocprof}; shift || true
args=${*:---steps 3 --warmup 1 --no-cpu-baseline --no-more}
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- python $repo/bench.py $args > /tmp/p_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -- python $repo/bench.py $args > /tmp/p_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -- python $repo/bench.py $args > /tmp/p_write.log 2>&1
cd $repo
{ echo "### p_stats: rocprofv3 --kernel-trace --stats -- python bench.py $args"; python tools/rocprof_summary.py /tmp/p_stats/*/*_results.db; grep '^{' /tmp/p_stats.log | tail -1 | cut -c1-400; } > ${out}_kernel_stats.txt 2>&1
{ echo "### p_fetch"; python tools/rocprof_summary.py /tmp/p_fetch/*/*_results.db; echo "### p_write"; python tools/rocprof_summary.py /tmp/p_write/*/*_results.db; } > ${out}_pmc_fetch_write.txt 2>&1
tail -n 40 ${out}_kernel_stats.txt
