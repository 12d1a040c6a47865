#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=20 > gpurun_out/r02b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b_pytest.log
tail -60 gpurun_out/r02b_pytest.log
tools/probe_f64_coexec > gpurun_out/r02b_probe_coexec.log 2>&1; cat gpurun_out/r02b_probe_coexec.log
python tools/time_predict.py > gpurun_out/r02b_time_predict.log 2>&1; tail -15 gpurun_out/r02b_time_predict.log
