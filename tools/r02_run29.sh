cd $GRAFT_REPO_ROOT
O=gpurun_out/r02z; mkdir -p $O
export GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_trace.so
timeout 200 python tools/trace_tiles.py 28 512x32 2 > $O/trace_p3_new.txt 2>&1
cat $O/trace_p3_new.txt
