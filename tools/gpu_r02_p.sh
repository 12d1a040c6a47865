#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02p_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02p_pytest.log; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r02p_pytest.log | tail -4
for c in c0 c2 c3; do python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], round(d['ms_per_step'],3), d['stage_ms_per_step'])"; done
