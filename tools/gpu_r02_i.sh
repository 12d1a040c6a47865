#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_primal.py -m gpu -q -x -k "bench or compressed or rccl" > gpurun_out/r02i_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02i_pytest.log; tail -6 gpurun_out/r02i_pytest.log
python __graft_entry__.py smoke 2>&1 | tail -3
