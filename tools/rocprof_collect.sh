#!/bin/bash
# Evidence run (round 6 layout) (GPU box): for keys-only, (u32,u32) and (u32,u64) pairs at 2^28 — rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE
# in runs of their own (counters never together with a trace domain other than --kernel-trace) — of `python bench.py --steps 3 --warmup 1
# --no-cpu-baseline --no-more [--pairs N]`.  usage: tools/rocprof_collect.sh OUT_DIR   -> OUT_DIR/{keys,pairs4,pairs8}_{stats,fetch,write}.txt
set -u
out=${1:-gpurun_out/r06_rocprof}
repo=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in keys pairs4 pairs8; do
  case $cfg in keys) extra="";; pairs4) extra="--pairs 4";; pairs8) extra="--pairs 8";; esac
  args="--steps 3 --warmup 1 --no-cpu-baseline --no-more $extra"
  rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
  rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- python $repo/bench.py $args > /tmp/p_stats.log 2>&1
  if [ -z "${ONLY_STATS:-}" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -- python $repo/bench.py $args > /tmp/p_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -- python $repo/bench.py $args > /tmp/p_write.log 2>&1
  fi
  { echo "### rocprofv3 --kernel-trace --stats -- python bench.py $args"; python $repo/tools/rocprof_summary.py /tmp/p_stats/*/*_results.db; grep '^{' /tmp/p_stats.log | tail -1 | cut -c1-600; } > $repo/$out/${cfg}_stats.txt 2>&1
  [ -n "${ONLY_STATS:-}" ] && continue
  { echo "### rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py $args"; python $repo/tools/rocprof_summary.py /tmp/p_fetch/*/*_results.db; } > $repo/$out/${cfg}_fetch.txt 2>&1
  { echo "### rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python bench.py $args"; python $repo/tools/rocprof_summary.py /tmp/p_write/*/*_results.db; } > $repo/$out/${cfg}_write.txt 2>&1
done
cd $repo
head -12 $out/keys_stats.txt
