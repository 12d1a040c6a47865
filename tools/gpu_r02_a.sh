#!/bin/bash
# Round-2 first GPU pass: full GPU test suite, bench c3 / c2, RCCL world-1 path.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a_pytest.log
tail -30 gpurun_out/r02a_pytest.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r02a_bench_c3.json 2> gpurun_out/r02a_bench_c3.err; echo "bench c3 rc=$?"
NLS_BENCH_FORCE_COMM=1 python bench.py --config c3e --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02a_bench_c3e_comm.json 2> gpurun_out/r02a_bench_c3e_comm.err; echo "bench c3e comm rc=$?"
python bench.py --config c2 --steps 10 --warmup 2 > gpurun_out/r02a_bench_c2.json 2> gpurun_out/r02a_bench_c2.err; echo "bench c2 rc=$?"
tail -c 1500 gpurun_out/r02a_bench_c3.json; tail -c 600 gpurun_out/r02a_bench_c3.err
tail -c 800 gpurun_out/r02a_bench_c3e_comm.json; tail -c 600 gpurun_out/r02a_bench_c3e_comm.err
tail -c 1500 gpurun_out/r02a_bench_c2.json; tail -c 600 gpurun_out/r02a_bench_c2.err
