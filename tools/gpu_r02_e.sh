#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r02e_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02e_pytest.log
tail -12 gpurun_out/r02e_pytest.log
python bench.py --steps 5 --warmup 1 > gpurun_out/r02e_bench_c3.json 2> gpurun_out/r02e_bench_c3.err; echo "bench c3 rc=$?"; tail -c 2500 gpurun_out/r02e_bench_c3.json
NLS_SWEEP_DIRECT=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02e_bench_c3_direct_sweep.json 2>/dev/null; tail -c 400 gpurun_out/r02e_bench_c3_direct_sweep.json
python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02e_bench_c2.json 2>/dev/null; tail -c 500 gpurun_out/r02e_bench_c2.json
python bench.py --config c3e --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02e_bench_c3e.json 2>/dev/null; tail -c 500 gpurun_out/r02e_bench_c3e.json
python tools/time_dual.py > gpurun_out/r02e_dual_c4.log 2>&1; tail -5 gpurun_out/r02e_dual_c4.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02e_prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02e_prof_bench.json 2> gpurun_out/r02e_prof_bench.err; echo "prof rc=$?"
find gpurun_out/r02e_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02e_c3_kernel_stats.csv; head -25 gpurun_out/r02e_c3_kernel_stats.csv
rm -rf gpurun_out/r02e_prof
