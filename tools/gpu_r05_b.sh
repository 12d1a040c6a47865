# Round 5, GPU pass B: one-launch tridiagonalisation, shared-column Cholesky panel, streaming small-G sweep - tests first (under timeout: the
# hand-off must not be able to hold the box), then timings.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_evd.py -x -q -m gpu > gpurun_out/r05b_evd.log 2>&1; echo "evd rc=$?"; tail -5 gpurun_out/r05b_evd.log
timeout 1200 python -m pytest tests/test_gpu_primal.py tests/test_gpu_dual.py tests/test_gpu_stedc.py -x -q -m gpu > gpurun_out/r05b_primal.log 2>&1; echo "primal rc=$?"; tail -5 gpurun_out/r05b_primal.log
for L in 2 1; do
  NLS_TRD_LAUNCHES=$L timeout 600 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > gpurun_out/r05b_bench_c2_L$L.json 2> gpurun_out/r05b_bench_c2_L$L.err; echo "c2 L=$L rc=$?"
  NLS_TRD_LAUNCHES=$L timeout 600 python bench.py --config c3e --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r05b_bench_c3e_L$L.json 2> gpurun_out/r05b_bench_c3e_L$L.err; echo "c3e L=$L rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05b_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],2), d["stage_ms_per_step"], d.get("evd_stage_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
