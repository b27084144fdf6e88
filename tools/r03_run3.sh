cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/$O/counters.txt
