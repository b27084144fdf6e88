set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
timeout 200 python tools/writeback_experiment.py > $O/writeback.txt 2>&1; cat $O/writeback.txt
