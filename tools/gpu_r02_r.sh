#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_primal.py tests/test_gpu_fullsize.py -m gpu -q -x -k "featuremap or predict or fit_matches or c3" > gpurun_out/r02r_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r02r_pytest.log | tail -1
python tools/time_predict.py 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['roofline_k1']['achieved'], d['stage_ms_per_step'])"
