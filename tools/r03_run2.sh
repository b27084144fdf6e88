set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_2; mkdir -p $O
export R03_MODES_ONLY=1
export GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp1024.so
R03_STRUCTURED=1 R03_MODES=0,768,1792,1024 timeout 300 python tools/r03_ablate.py 28 3 0 > $O/structured.txt 2>&1; cat $O/structured.txt
for c in 4 8 16 32; do
  export GPUSORT_LIB=$PWD/gpusorting_amd/lib/libgpusort_exp1024_nch$c.so
  R03_MODES=0,768,1792 timeout 300 python tools/r03_ablate.py 28 3 0 > $O/nch$c.txt 2>&1; cat $O/nch$c.txt
done
