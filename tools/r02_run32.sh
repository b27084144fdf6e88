cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
for W in 0 32 64 96 128 192 256 384; do GPUSORT_HIST_BLOCKS=$W timeout 120 python tools/hist_blocks_sweep.py >> $O/hist_blocks.txt 2>&1; done
grep blocks $O/hist_blocks.txt | sort -k2,2 -s
