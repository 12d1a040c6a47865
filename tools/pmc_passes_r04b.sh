# Round-4 counter passes, second call: k_rotate3 in the default (8-row XCD patch) order.  The first call (tools/pmc_passes_r04.sh) ran these
# three passes FIRST on a fresh box and all three timed out with empty logs while the plain-order passes right after them succeeded; here one
# plain pass warms the box first and the patch order is also requested explicitly, to tell a start-up stall from a kernel problem.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
g++ -O2 tools/nls_cbench.cpp -Iinclude -Lneo_ls_svm_amd -lneolssvm_hip -Wl,-rpath,$PWD/neo_ls_svm_amd -o /tmp/nls_cbench || exit 1
mkdir -p gpurun_out/r04
LOG=gpurun_out/r04/pmc_passes_b.log; : > $LOG
( export NLS_ROT_PATCH=0x0; timeout 120 /tmp/nls_cbench 333440 128 4096 1024 rotate 1 ) >> $LOG 2>&1; echo "warm plain rc=$?" >> $LOG
( timeout 120 /tmp/nls_cbench 333440 128 4096 1024 rotate 1 ) >> $LOG 2>&1; echo "warm default rc=$?" >> $LOG
run() {  # tag, what, env...
  tag=$1; what=$2; shift 2
  i=0
  for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    ( export "$@"; timeout 240 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcR2_${tag}_$i -- /tmp/nls_cbench 333440 128 4096 1024 $what 1 > gpurun_out/pmcR2_${tag}_$i.log 2>&1 ); echo "$tag $i rc=$?" >> $LOG
  done
}
run rot_default rotate NLS_DUMMY=1
run rot_p8x5 rotate NLS_ROT_PATCH=8x5
python tools/pmc_summarise.py gpurun_out > gpurun_out/r04_pmc_passes_b.json 2>> $LOG
find gpurun_out -path "*pmcR2_*" -name "*.csv" -size +2M -delete
tail -20 $LOG
