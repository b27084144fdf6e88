cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O; rm -f $O/hist_unroll.txt
for rep in 1 2; do for L in libgpusort.so libgpusort_hu2.so libgpusort_hu8.so; do echo "lib=$L" >> $O/hist_unroll.txt; GPUSORT_LIB=$PWD/gpusorting_amd/lib/$L timeout 120 python tools/hist_blocks_sweep.py 22,24,26,28 >> $O/hist_unroll.txt 2>&1; done; done
cat $O/hist_unroll.txt
