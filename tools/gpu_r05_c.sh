# Round 5, GPU pass C: Gram tile orders (time + traffic past L2), the rotation's FETCH_SIZE pass that a cold profiler lost in pass A, bench lines after
# the end-to-end fix (c2), the dual path (c4), one rank's share (c3e), the 16 x 32 grid behind the C ABI with the streaming small-G sweep (c5).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_primal.py tests/test_gpu_dual.py -x -q -m gpu > gpurun_out/r05c_primal.log 2>&1; echo "primal rc=$?"; tail -3 gpurun_out/r05c_primal.log
g++ -O2 tools/nls_cbench.cpp -Iinclude -Lneo_ls_svm_amd -lneolssvm_hip -Wl,-rpath,$PWD/neo_ls_svm_amd -o /tmp/nls_cbench || exit 1
for o in plain contiguous patch; do
  echo "== gram order $o"; NLS_GRAM_ORDER=$o /tmp/nls_cbench 333440 128 4096 1024 gram 4 2>&1 | tail -4
done > gpurun_out/r05c_gram_orders.log 2>&1
cat gpurun_out/r05c_gram_orders.log
rm -rf gpurun_out/pmcR2_*
# a throw-away profiled run first: the first rocprofv3 --pmc pass on a fresh box stalls at start-up (r04, r05 pass A)
( timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcR2_warm_1 -- /tmp/nls_cbench 8192 128 4096 1024 rotate 1 > gpurun_out/pmcR2_warm_1.log 2>&1 ); echo "warm rc=$?"
run() {  # tag, what, env...
  tag=$1; what=$2; shift 2
  i=0
  for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    ( export "$@"; timeout 300 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcR2_${tag}_$i -- /tmp/nls_cbench 333440 128 4096 1024 $what 1 > gpurun_out/pmcR2_${tag}_$i.log 2>&1 ); echo "$tag $i rc=$?"
  done
}
run rot_default rotate NLS_DUMMY=1
run gram_patch gram NLS_GRAM_ORDER=patch
rm -rf gpurun_out/pmcR2_warm_1
python tools/pmc_summarise.py gpurun_out > gpurun_out/r05c_pmc_passes.json; cat gpurun_out/r05c_pmc_passes.json
find gpurun_out -path "*pmcR2_*" -name "*.csv" -size +2M -delete
python bench.py --config c2 --steps 20 --warmup 3 > gpurun_out/r05c_bench_c2.json 2> gpurun_out/r05c_bench_c2.err; echo "c2 rc=$?"
python bench.py --config c3e --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05c_bench_c3e.json 2> gpurun_out/r05c_bench_c3e.err; echo "c3e rc=$?"
python bench.py --config c4 --steps 10 --warmup 2 > gpurun_out/r05c_bench_c4.json 2> gpurun_out/r05c_bench_c4.err; echo "c4 rc=$?"
python bench.py --config c5 --steps 2 --warmup 1 > gpurun_out/r05c_bench_c5.json 2> gpurun_out/r05c_bench_c5.err; echo "c5 rc=$?"
NLS_SWEEP_SMALL=0 python bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05c_bench_c5_bigtile.json 2> gpurun_out/r05c_bench_c5_bigtile.err; echo "c5 big tile rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],4), round(d["ms_per_step"],2), d["stage_ms_per_step"], d.get("value_end_to_end"), (d.get("end_to_end") or {}).get("stage_seconds"))
    except Exception as e:
        print(f, "unreadable", e)
PY
