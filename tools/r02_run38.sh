cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final2; mkdir -p $O
timeout 300 python tools/placement_experiment.py 8 > $O/placement_one_arena.txt 2>&1
timeout 300 python tools/placement_experiment.py 8 >> $O/placement_one_arena.txt 2>&1
cat $O/placement_one_arena.txt
