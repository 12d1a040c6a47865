#!/bin/bash
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_twostage.py -x -q -m gpu 2>&1 | tail -3
NLS_Q2_FORM=wave timeout 900 python -m pytest tests/test_gpu_twostage.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/bench_c4_run13_$i.json 2> gpurun_out/r04/bench_c4_run13_$i.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04/bench_c4_run13_$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("evd_stage_ms"))
PY
done
