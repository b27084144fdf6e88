cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_4; mkdir -p $O
export GPUSORT_LIB=$R/gpusorting_amd/lib/libgpusort_exp1024.so
export R03_MODES_ONLY=1 R03_NOPRIME=1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_REQ_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_BUSY_sum"; do
  i=$((i+1))
  for m in 0 1024; do
    rm -rf /tmp/pmc_$i_$m
    R03_MODES=$m timeout 200 rocprofv3 --pmc $grp -d /tmp/pmc_${i}_$m -- python $R/tools/r03_ablate.py 28 2 0 > $O/run_${i}_$m.log 2>&1
    echo "=== group $i mode $m: $grp" >> $O/pmc.txt
    python $R/tools/rocprof_summary.py $(find /tmp/pmc_${i}_$m -name "*.db" | head -1) 2>&1 | grep -E "digit_binning|global_hist" | grep -v "^--" >> $O/pmc.txt
  done
done
cat $O/pmc.txt
