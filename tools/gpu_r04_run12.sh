#!/bin/bash
# Wave-per-block second back-transformation: stage tests, then the c4 line in both forms.
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_twostage.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r04/pytest_run12.log; cat gpurun_out/r04/pytest_run12.log
for form in wave team; do
  NLS_Q2_FORM=$form timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/bench_c4_q2$form.json 2> gpurun_out/r04/bench_c4_q2$form.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04/bench_c4_q2$form.json").read().strip().splitlines()[-1])
print("$form", d["value"], d["ms_per_step"], d.get("evd_stage_ms"))
PY
done
