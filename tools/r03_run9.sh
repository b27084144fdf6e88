cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_9; mkdir -p $O
export R03_MODES_ONLY=1 R03_NOPRIME=1 GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp2048.so
for p in 0 2 4; do
GPUSORT_SHAPE=512x16 R03_MODES=0,2048,0,2048 timeout 300 python tools/r03_ablate.py 28 3 $p > $O/count_512x16_p$p.txt 2>&1; cat $O/count_512x16_p$p.txt
done
