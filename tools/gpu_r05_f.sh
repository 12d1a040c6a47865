# Round 5, GPU pass F: the pre-step's two host sorts on the device (nls_rank_codes, nls_bin_stats_labels) - tests, then the end-to-end legs.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_estimator.py tests/test_gpu_baseline_sizes.py tests/test_gpu_group.py -x -q -m gpu > gpurun_out/r05f_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05f_tests.log
python bench.py --config c2 --steps 20 --warmup 3 > gpurun_out/r05f_bench_c2.json 2> gpurun_out/r05f_bench_c2.err; echo "c2 rc=$?"
python bench.py > gpurun_out/r05f_bench_c3.json 2> gpurun_out/r05f_bench_c3.err; echo "c3 rc=$?"
python tools/profile_fit.py 1000000 128 4096 > gpurun_out/r05f_profile_fit_c3.log 2>&1; echo "profile c3 rc=$?"
python tools/profile_fit.py 100000 64 1024 > gpurun_out/r05f_profile_fit_c2.log 2>&1; echo "profile c2 rc=$?"
python - <<'PY'
import json
for f in ("r05f_bench_c2","r05f_bench_c3"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"],4), round(d["ms_per_step"],2), d.get("value_end_to_end"), d["end_to_end"]["stage_seconds"])
PY
head -30 gpurun_out/r05f_profile_fit_c3.log
