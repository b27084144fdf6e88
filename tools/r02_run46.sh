cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final3; mkdir -p $O
timeout 600 ./build/gpusorting_main 28 100 > $O/gpusorting_main.txt 2>&1; tail -12 $O/gpusorting_main.txt
timeout 300 ./build/rocprim_compare > $O/rocprim.txt 2>&1; tail -8 $O/rocprim.txt
timeout 300 ./build/gpusorting_d3d12_main supertest > $O/d3d12.txt 2>&1; tail -3 $O/d3d12.txt
