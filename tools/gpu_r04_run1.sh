#!/bin/bash
# round 4, GPU run 1: tests, bench lines, two-context experiment
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_gpu.log
tail -5 gpurun_out/r04/pytest_gpu.log
timeout 300 python bench.py --config c2 --steps 10 --warmup 2 > gpurun_out/r04/bench_c2.json 2> gpurun_out/r04/bench_c2.err; echo "c2 rc $?"
timeout 600 python bench.py > gpurun_out/r04/bench_c3.json 2> gpurun_out/r04/bench_c3.err; echo "c3 rc $?"
timeout 400 python bench.py --config c4 --steps 5 --warmup 1 > gpurun_out/r04/bench_c4.json 2> gpurun_out/r04/bench_c4.err; echo "c4 rc $?"
timeout 300 python bench.py --config c3e --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/bench_c3e.json 2> gpurun_out/r04/bench_c3e.err; echo "c3e rc $?"
timeout 400 python tools/dev_two_context.py 20000 8 > gpurun_out/r04/two_ctx_20000.log 2>&1; echo "two ctx small rc $?"
timeout 400 python tools/dev_two_context.py 125000 8 > gpurun_out/r04/two_ctx_125000.log 2>&1; echo "two ctx 125k rc $?"
timeout 600 python tools/dev_two_context.py 1000000 3 > gpurun_out/r04/two_ctx_1e6.log 2>&1; echo "two ctx 1e6 rc $?"
tail -3 gpurun_out/r04/two_ctx_*.log
