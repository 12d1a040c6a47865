#!/bin/bash
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dual.py tests/test_gpu_two_contexts.py tests/test_gpu_primal.py -m gpu -x -q > gpurun_out/r04/pytest_run4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_run4.log
tail -5 gpurun_out/r04/pytest_run4.log
timeout 300 python tools/dev_c2_idle.py > gpurun_out/r04/c2_idle.log 2>&1; cat gpurun_out/r04/c2_idle.log
timeout 300 python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_c2_d.json 2> gpurun_out/r04/bench_c2_d.err; echo "c2 rc $?"
timeout 300 python bench.py --config c3e --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/bench_c3e_d.json 2> gpurun_out/r04/bench_c3e_d.err; echo "c3e rc $?"
python - <<'PY'
import json
for c in ("c2_d","c3e_d"):
    d=json.loads(open(f"gpurun_out/r04/bench_{c}.json").read().strip().splitlines()[-1])
    print(c, d["value"], d["stage_ms_per_step"], d.get("end_to_end"))
PY
