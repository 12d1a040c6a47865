#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r02final2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02final2_pytest.log; tail -12 gpurun_out/r02final2_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 5 --warmup 2 > gpurun_out/r02final2_bench.json 2> gpurun_out/r02final2_bench.err; echo "bench rc=$?"; wc -l gpurun_out/r02final2_bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02final2_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_k1']['frac'], d['stage_ms_per_step'], d['cpu_baseline']['gpu_over_cpu'])
PY
python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline | tail -c 330
python tools/time_dual.py | tail -3
