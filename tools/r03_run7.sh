cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_7; mkdir -p $O
timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_pf0.so gpusorting_amd/lib/libgpusort_min_nt.so gpusorting_amd/lib/libgpusort_min_ntpf.so --rounds 3 --vb 0 > $O/ab.txt 2>&1; cat $O/ab.txt
GPUSORT_HIST_BLOCKS=512 timeout 300 python tools/ab.py gpusorting_amd/lib/libgpusort_min_pf0.so gpusorting_amd/lib/libgpusort_min_nt.so gpusorting_amd/lib/libgpusort_min_ntpf.so --rounds 2 --vb 0 > $O/ab512.txt 2>&1; cat $O/ab512.txt
