set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
GPUSORT_RANK=0 timeout 300 python tools/entropy_breakdown.py 28 4 0 > $O/entropy_rank0.txt 2>&1
timeout 300 python tools/entropy_breakdown.py 28 4 0 > $O/entropy_rank1.txt 2>&1
cat $O/entropy_rank0.txt $O/entropy_rank1.txt
