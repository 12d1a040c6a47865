# Round 5, GPU pass D: the Gram kernel's XCD patch order with the patch list balanced over the XCDs - bit identity, time, traffic.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_primal.py -x -q -m gpu -k "gram" > gpurun_out/r05d_gram_tests.log 2>&1; echo "gram tests rc=$?"; tail -3 gpurun_out/r05d_gram_tests.log
g++ -O2 tools/nls_cbench.cpp -Iinclude -Lneo_ls_svm_amd -lneolssvm_hip -Wl,-rpath,$PWD/neo_ls_svm_amd -o /tmp/nls_cbench || exit 1
for o in plain patch contiguous plain patch; do
  echo "== gram order $o"; NLS_GRAM_ORDER=$o /tmp/nls_cbench 333440 128 4096 1024 gram 4 2>&1 | tail -3
done > gpurun_out/r05d_gram_orders.log 2>&1
cat gpurun_out/r05d_gram_orders.log
rm -rf gpurun_out/pmcR2_*
( timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcR2_warm_1 -- /tmp/nls_cbench 8192 128 4096 1024 rotate 1 > gpurun_out/pmcR2_warm_1.log 2>&1 ); echo "warm rc=$?"
run() {
  tag=$1; what=$2; shift 2
  i=0
  for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    ( export "$@"; timeout 300 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcR2_${tag}_$i -- /tmp/nls_cbench 333440 128 4096 1024 $what 1 > gpurun_out/pmcR2_${tag}_$i.log 2>&1 ); echo "$tag $i rc=$?"
  done
}
run gram_patch gram NLS_GRAM_ORDER=patch
run gram_plain gram NLS_GRAM_ORDER=plain
rm -rf gpurun_out/pmcR2_warm_1
python tools/pmc_summarise.py gpurun_out > gpurun_out/r05d_pmc_passes.json; cat gpurun_out/r05d_pmc_passes.json | grep -A9 k_gram3
find gpurun_out -path "*pmcR2_*" -name "*.csv" -size +2M -delete
