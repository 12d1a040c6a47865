# Round 5, GPU pass G: 32-column slabs in the second back-transformation (k_q2_apply_packed<.., NCT = 2>): bit identity, then time.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_twostage.py -x -q -m gpu > gpurun_out/r05g_twostage.log 2>&1; echo "twostage rc=$?"; tail -3 gpurun_out/r05g_twostage.log
for nct in 1 2; do
  NLS_Q2_NCT=$nct timeout 600 python bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r05g_bench_c4_nct$nct.json 2> gpurun_out/r05g_bench_c4_nct$nct.err; echo "c4 nct=$nct rc=$?"
done
timeout 600 python bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r05g_bench_c4_default.json 2> gpurun_out/r05g_bench_c4_default.err; echo "c4 default rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05g_bench_c4_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],2), d.get("evd_stage_ms"))
    except Exception as e:
        print(f, "unreadable", e, open(f.replace(".json",".err")).read()[-1500:])
PY
