cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O; rm -f $O/hist_blocks_large.txt
for rep in 1 2; do for W in 0 256 320 384 448; do GPUSORT_HIST_BLOCKS=$W timeout 120 python tools/hist_blocks_sweep.py 27,28 >> $O/hist_blocks_large.txt 2>&1; done; done
grep blocks $O/hist_blocks_large.txt | sort -k2,2 -s
