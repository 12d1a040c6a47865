# Round 5, GPU pass I: counter evidence for the streaming small-G sweep kernel (k_sweep_small, the 32-point grid of config 5): bytes past L2 and time
# of one launch over 5e5 rows at D = 4096 (the fit of the native driver with G = 32 takes the direct sweep).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
g++ -O2 tools/nls_cbench.cpp -Iinclude -Lneo_ls_svm_amd -lneolssvm_hip -Wl,-rpath,$PWD/neo_ls_svm_amd -o /tmp/nls_cbench || exit 1
/tmp/nls_cbench 500000 128 4096 32 fit 2 2>&1 | tail -2
rm -rf gpurun_out/pmcR2_*
( timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcR2_warm_1 -- /tmp/nls_cbench 8192 128 4096 32 fit 1 > gpurun_out/pmcR2_warm_1.log 2>&1 ); echo "warm rc=$?"
i=0
for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  ( timeout 300 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcR2_sweep32_$i -- /tmp/nls_cbench 500000 128 4096 32 fit 1 > gpurun_out/pmcR2_sweep32_$i.log 2>&1 ); echo "sweep32 $i rc=$?"
done
rm -rf gpurun_out/pmcR2_warm_1
python tools/pmc_summarise.py gpurun_out > gpurun_out/r05i_pmc_passes.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05i_pmc_passes.json"))
for tag,ks in d.items():
    for k,e in ks.items():
        if k.startswith("k_sweep"): print(tag,k,e)
PY
find gpurun_out -path "*pmcR2_*" -name "*.csv" -size +2M -delete
