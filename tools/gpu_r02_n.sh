#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_evd.py -m gpu -q -x > gpurun_out/r02n_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02n_pytest.log; tail -8 gpurun_out/r02n_pytest.log
for cfg in "0 32" "2304 32" "2304 16" "2304 24"; do set -- $cfg; echo "== PERSIST_MAX=$1 WG=$2 (one XCD)";
  for n in 257 513 1025 1537 2049; do NLS_TRD_PERSIST_MAX=$1 NLS_TRD_PERSIST_WG=$2 timeout 120 python tools/time_evd.py $n c 3 | tail -1; done
  NLS_TRD_PERSIST_MAX=$1 NLS_TRD_PERSIST_WG=$2 timeout 120 python tools/time_evd.py 1000 r 3 | tail -1
  NLS_TRD_PERSIST_MAX=$1 NLS_TRD_PERSIST_WG=$2 timeout 300 python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline | tail -c 330
  NLS_TRD_PERSIST_MAX=$1 NLS_TRD_PERSIST_WG=$2 timeout 300 python bench.py --config c0 --steps 10 --warmup 3 --no-cpu-baseline | tail -c 330
done > gpurun_out/r02n_persist.log 2>&1
cat gpurun_out/r02n_persist.log
