# Round-4 counter passes (one counter set per pass; kernel trace only, as gpurun requires) for the two MFMA kernels in their DEFAULT tile
# orders of this round - k_rotate3: padding-free 8-row XCD patch; k_gram3: contiguous run of the (split, half tile) list per XCD - beside the
# plain orders, plus the feature map.  333 440 rows (one launch), d = 128, D = 4096.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
g++ -O2 tools/nls_cbench.cpp -Iinclude -Lneo_ls_svm_amd -lneolssvm_hip -Wl,-rpath,$PWD/neo_ls_svm_amd -o /tmp/nls_cbench || exit 1
rm -rf gpurun_out/pmcR2_*
run() {  # tag, what, env...
  tag=$1; what=$2; shift 2
  i=0
  for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    env "$@" true
    ( export "$@"; timeout 180 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcR2_${tag}_$i -- /tmp/nls_cbench 333440 128 4096 1024 $what 1 > gpurun_out/pmcR2_${tag}_$i.log 2>&1 ); echo "$tag $i rc=$?"
  done
}
run rot_default rotate NLS_DUMMY=1
run rot_plain rotate NLS_ROT_PATCH=0x0
run gram_default gram NLS_DUMMY=1
run gram_plain gram NLS_GRAM_ORDER=plain
python tools/pmc_summarise.py gpurun_out > gpurun_out/r04_pmc_passes.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_pmc_passes.json"))
for tag,ks in d.items():
    for k,e in ks.items():
        print(tag, k, {a:(round(v,3) if isinstance(v,float) and v<100 else (round(v/1e9,2) if isinstance(v,float) else v)) for a,v in e.items()})
rot=d["rot_default"]["k_rotate3"]; fm=d["rot_default"]["k_featuremap"]; sp=d["rot_default"].get("k_shift_pad",{})
rows=333440
out={"k_rotate3":{"D":4096,"d":128,"rows_per_launch":rows,"fetch_bytes_x2":rot["fetch_bytes_x2_per_launch"],"write_bytes":rot["write_bytes_per_launch"],"l2_hit":rot.get("l2_hit"),"ms":rot.get("avg_ms"),"order":"default (8-row XCD patch)"},
     "k_rotate3_plain":d["rot_plain"]["k_rotate3"],
     "k_gram3":dict(d["gram_default"]["k_gram3"],order="default (XCD-contiguous)"),
     "k_gram3_plain":d["gram_plain"]["k_gram3"],
     "k_featuremap":{"D":4096,"d":128,"rows_per_launch":rows,"fetch_bytes_x2":fm["fetch_bytes_x2_per_launch"],"write_bytes":fm["write_bytes_per_launch"],"ms":fm.get("avg_ms"),
                     "hbm_bytes_per_row":(fm["fetch_bytes_x2_per_launch"]+fm["write_bytes_per_launch"]+sp.get("fetch_bytes_x2_per_launch",0)+sp.get("write_bytes_per_launch",0))/rows}}
json.dump(out,open("gpurun_out/r04_pmc_summary.json","w"),indent=1)
PY
find gpurun_out -path "*pmcR2_*" -name "*.csv" -size +2M -delete
