set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02v; mkdir -p $O
timeout 300 python tools/pass_experiment.py > $O/pass_experiment.txt 2>&1; cat $O/pass_experiment.txt
