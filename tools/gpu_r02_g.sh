#!/bin/bash
mkdir -p gpurun_out
python tools/numerics_report.py > gpurun_out/r02g_numerics.txt 2>&1; cat gpurun_out/r02g_numerics.txt
python bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02g_bench_c5.json 2> gpurun_out/r02g_bench_c5.err; echo "c5 rc=$?"; tail -c 1200 gpurun_out/r02g_bench_c5.json; tail -3 gpurun_out/r02g_bench_c5.err
python tools/profile_fit.py > gpurun_out/r02g_profile_fit.log 2>&1; tail -12 gpurun_out/r02g_profile_fit.log
