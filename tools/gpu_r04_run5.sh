#!/bin/bash
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_stedc.py -m gpu -x -q > gpurun_out/r04/pytest_run5a.log 2>&1; echo "stedc pytest rc $?" >> gpurun_out/r04/pytest_run5a.log
tail -15 gpurun_out/r04/pytest_run5a.log
timeout 900 python -m pytest tests/test_gpu_dual.py tests/test_gpu_two_contexts.py tests/test_gpu_primal.py tests/test_gpu_evd.py tests/test_gpu_twostage.py -m gpu -x -q > gpurun_out/r04/pytest_run5b.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_run5b.log
tail -8 gpurun_out/r04/pytest_run5b.log
for c in c2 c3e c4; do
timeout 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/bench_${c}_e.json 2> gpurun_out/r04/bench_${c}_e.err; echo "$c rc $?"
done
python - <<'PY'
import json
for c in ("c2_e","c3e_e","c4_e"):
    try:
        d=json.loads(open(f"gpurun_out/r04/bench_{c}.json").read().strip().splitlines()[-1])
        print(c, d["value"], d["stage_ms_per_step"], d.get("evd_stage_ms"))
    except Exception as e:
        print(c, "ERR", e); print(open(f"gpurun_out/r04/bench_{c}.err").read()[-800:])
PY
