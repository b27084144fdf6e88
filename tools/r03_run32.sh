cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_32; mkdir -p $O
for lg in 23 24; do
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_trace.so GPUSORT_MID_PATH=0 timeout 300 python tools/trace_tiles.py $lg 512x16 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
done
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_trace.so GPUSORT_MID_PATH=0 timeout 300 python tools/trace_tiles.py 23 512x32 2>&1 | grep -v amdgpu.ids >> $O/trace.txt
cat $O/trace.txt
