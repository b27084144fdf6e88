cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02z; timeout 60 ./build/ticket_microbench > gpurun_out/r02z/ticket_microbench.txt 2>&1; cat gpurun_out/r02z/ticket_microbench.txt
