cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for what in rotate gram; do
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmcL_${what} -- ./tools/nls_cbench 131072 128 4096 1024 $what 1 > gpurun_out/pmcL_${what}.log 2>&1; echo "$what rc=$?"
done
