#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_dual.py tests/test_gpu_estimator.py tests/test_gpu_baseline_sizes.py tests/test_gpu_fullsize.py -m gpu -q -x -k "rccl_comm or dual or c4" > gpurun_out/r02j_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02j_pytest.log; tail -6 gpurun_out/r02j_pytest.log
python tools/time_dual.py > gpurun_out/r02j_dual_c4.log 2>&1; tail -4 gpurun_out/r02j_dual_c4.log
