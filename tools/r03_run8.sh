cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_8; mkdir -p $O
GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_min.so timeout 600 python tools/r03_poschain.py 28 3 > $O/poschain.txt 2>&1; cat $O/poschain.txt
export R03_MODES_ONLY=1 R03_NOPRIME=1 GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp2048.so
for p in 0 2; do
GPUSORT_SHAPE=512x16 R03_MODES=0,2048,0,2048 timeout 300 python tools/r03_ablate.py 28 3 $p > $O/count_512x16_p$p.txt 2>&1; cat $O/count_512x16_p$p.txt
R03_MODES=0,2048,0,2048 timeout 300 python tools/r03_ablate.py 28 3 $p > $O/count_512x32_p$p.txt 2>&1; cat $O/count_512x32_p$p.txt
done
