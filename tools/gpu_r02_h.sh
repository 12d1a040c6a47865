#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r02h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02h_pytest.log; tail -4 gpurun_out/r02h_pytest.log
python bench.py --config c5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02h_bench_c5.json 2> gpurun_out/r02h_bench_c5.err; echo "c5 rc=$?"; tail -c 700 gpurun_out/r02h_bench_c5.json; tail -3 gpurun_out/r02h_bench_c5.err
python tools/profile_fit.py > gpurun_out/r02h_profile_fit.log 2>&1; head -12 gpurun_out/r02h_profile_fit.log
