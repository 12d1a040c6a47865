cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for what in rotate gram; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcD_${what}_$i -- ./tools/nls_cbench 333440 128 4096 1024 $what 1 > gpurun_out/pmcD_${what}_$i.log 2>&1; echo "$what $i rc=$?"
  done
done
