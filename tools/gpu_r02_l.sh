#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_evd.py tests/test_gpu_baseline_sizes.py -m gpu -q -x -k "trid or eigh or panel or rank2k or large_n" > gpurun_out/r02l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02l_pytest.log; tail -5 gpurun_out/r02l_pytest.log
for g in 0 1; do echo "== NLS_TRD_GRAPH=$g";
  NLS_TRD_GRAPH=$g python tools/time_evd.py 1025 c 4; NLS_TRD_GRAPH=$g python tools/time_evd.py 4097 c 4; NLS_TRD_GRAPH=$g python tools/time_evd.py 10000 r 4
  NLS_TRD_GRAPH=$g python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline | tail -c 330
  NLS_TRD_GRAPH=$g python bench.py --config c3e --steps 3 --warmup 2 --no-cpu-baseline | tail -c 330
done > gpurun_out/r02l_graph.log 2>&1
cat gpurun_out/r02l_graph.log
