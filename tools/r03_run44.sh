cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_44; mkdir -p $O
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp5120f.so timeout 300 python tools/r03_mid_phases.py 16 22 2>&1 | grep -v amdgpu.ids > $O/mid_phases.txt
cat $O/mid_phases.txt
