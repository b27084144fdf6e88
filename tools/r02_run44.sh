cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final3; mkdir -p $O; rm -f $O/flush64.txt
for rep in 1 2; do for L in libgpusort.so libgpusort_flush64.so; do echo "lib=$L" >> $O/flush64.txt; GPUSORT_LIB=$PWD/gpusorting_amd/lib/$L timeout 120 python tools/hist_blocks_sweep.py 21,22,23,24,26,28 >> $O/flush64.txt 2>&1; done; done
cat $O/flush64.txt
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_flush64.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hist or golden or entropy" 2>&1 | grep -E "passed|failed"
