cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final2; mkdir -p $O; rm -f $O/crowded.txt
for rep in 1 2; do for L in libgpusort.so libgpusort_crowded.so; do GPUSORT_LIB=$PWD/gpusorting_amd/lib/$L timeout 300 python tools/entropy_breakdown.py 28 4 8 >> $O/crowded.txt 2>&1; done; done
grep -E "^lib|^vb" $O/crowded.txt
