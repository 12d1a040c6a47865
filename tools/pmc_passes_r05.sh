# Round-5 counter passes on the CURRENT build (one counter set per pass; kernel trace only, as gpurun requires; the native driver directly
# after "--"): the two MFMA kernels in their single-GPU default tile orders (k_rotate3: 8-row XCD patch; k_gram3: plain) and the Gram kernel's
# XCD-contiguous order (the default when ranks share the fabric).  Sets: traffic past L2, L2 hit rate, and the matrix-pipe occupancy
# (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE) the bench line carries as roofline.mfma_busy.  333 440 rows (one launch), d = 128, D = 4096.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
g++ -O2 tools/nls_cbench.cpp -Iinclude -Lneo_ls_svm_amd -lneolssvm_hip -Wl,-rpath,$PWD/neo_ls_svm_amd -o /tmp/nls_cbench || exit 1
rm -rf gpurun_out/pmcR2_*
/tmp/nls_cbench 8192 128 4096 1024 rotate 1 > /dev/null 2>&1   # page the libraries in before the first profiled pass (r04: it timed out on a cold box)
run() {  # tag, what, env...
  tag=$1; what=$2; shift 2
  i=0
  for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    ( export "$@"; timeout 300 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcR2_${tag}_$i -- /tmp/nls_cbench 333440 128 4096 1024 $what 1 > gpurun_out/pmcR2_${tag}_$i.log 2>&1 ); echo "$tag $i rc=$?"
  done
}
run rot_default rotate NLS_DUMMY=1
run gram_default gram NLS_DUMMY=1
run gram_contig gram NLS_GRAM_ORDER=contiguous
python tools/pmc_summarise.py gpurun_out > gpurun_out/r05_pmc_passes.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_pmc_passes.json"))
rows=333440
rot=d["rot_default"]["k_rotate3"]; fm=d["rot_default"]["k_featuremap"]; sp=d["rot_default"].get("k_shift_pad",{})
def pick(e, **kw):
    o={"rows_per_launch":rows,"D":4096,"d":128,"fetch_bytes_x2":e.get("fetch_bytes_x2_per_launch"),"write_bytes":e.get("write_bytes_per_launch"),
       "l2_hit":e.get("l2_hit"),"ms":e.get("avg_ms"),"mfma_busy":e.get("mfma_busy"),"clock_ghz":e.get("clock_ghz")}
    o.update(kw); return o
out={"_note":"rocprofv3 --kernel-trace --pmc, one counter set per pass (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES), tools/pmc_passes_r05.sh on the round-5 build; bytes past L2 = 2 x FETCH_SIZE (gfx950 wide reads, MI355X_MICROARCH.md) and WRITE_SIZE, KiB -> bytes; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 matrix pipes); 333 440 rows per launch, d = 128, D = 4096",
     "k_rotate3":pick(rot,order="default = 8-row XCD patch"),
     "k_gram3":pick(d["gram_default"]["k_gram3"],order="default on one GPU = plain"),
     "k_gram3_contiguous":pick(d["gram_contig"]["k_gram3"],order="XCD-contiguous (default with a communicator)"),
     "k_featuremap":{"D":4096,"d":128,"rows_per_launch":rows,"fetch_bytes_x2":fm.get("fetch_bytes_x2_per_launch"),"write_bytes":fm.get("write_bytes_per_launch"),"ms":fm.get("avg_ms"),
                     "hbm_bytes_per_row":(fm.get("fetch_bytes_x2_per_launch",0)+fm.get("write_bytes_per_launch",0)+sp.get("fetch_bytes_x2_per_launch",0)+sp.get("write_bytes_per_launch",0))/rows}}
json.dump(out,open("gpurun_out/r05_pmc_summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
find gpurun_out -path "*pmcR2_*" -name "*.csv" -size +2M -delete
