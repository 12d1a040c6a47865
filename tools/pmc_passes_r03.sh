# Round-3 counter passes for k_rotate3 (one counter set per pass; kernel trace only, as gpurun requires): XCD-patch shapes whose column
# count divides the 65 column tiles of D + 1 = 4097 (5, 13: no padding blocks) beside the plain order and the 4 x 8 patch of round 2.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in 0x0 4x8 4x5 8x5 4x13 2x13 8x13; do
  tag="p$p"
  i=0
  for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    NLS_ROT_PATCH=$p rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcR2_${tag}_$i -- ./tools/nls_cbench 333440 128 4096 1024 rotate 1 > gpurun_out/pmcR2_${tag}_$i.log 2>&1; echo "$tag $i rc=$?"
  done
done
python tools/pmc_summarise.py gpurun_out > gpurun_out/r03_pmc_rotate.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_pmc_rotate.json"))
for tag,ks in d.items():
    e=ks.get("k_rotate3")
    if e: print(tag, {k:(round(v,3) if isinstance(v,float) and v<100 else (round(v/1e9,1) if isinstance(v,float) else v)) for k,v in e.items()})
PY
find gpurun_out -path "*pmcR2_*" -name "*.csv" -size +2M -delete
