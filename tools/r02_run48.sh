cd $GRAFT_REPO_ROOT
O=gpurun_out/r02pipe; mkdir -p $O; rm -f $O/*.txt
for P in 0 1 0 1; do echo "GPUSORT_PIPE=$P" >> $O/pipe_ab.txt; GPUSORT_PIPE=$P timeout 200 python tools/entropy_breakdown.py 28 4 0 >> $O/pipe_ab.txt 2>&1; done
grep -E "PIPE|^vb" $O/pipe_ab.txt
GPUSORT_PIPE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "not fault" > $O/pytest_pipe.txt 2>&1; grep -E "passed|failed|rror" $O/pytest_pipe.txt | tail -3
