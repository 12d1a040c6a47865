# Round 5, closing pass on the final build: the whole GPU suite, smoke, the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r05_smoke.log
python bench.py > gpurun_out/r05_bench_c3_final.json 2> gpurun_out/r05_bench_c3_final.err; echo "c3 rc=$?"
python bench.py --config c2 --steps 20 --warmup 3 > gpurun_out/r05_bench_c2_final.json 2> gpurun_out/r05_bench_c2_final.err; echo "c2 rc=$?"
python - <<'PY'
import json
for f in ("r05_bench_c3_final","r05_bench_c2_final"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"],4), round(d["ms_per_step"],2), d.get("value_end_to_end"), d["end_to_end"]["stage_seconds"], d["cpu_baseline"].get("full_size_check",{}).get("extrapolation_over_measured"))
PY
