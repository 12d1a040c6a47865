#!/bin/bash
# round 4, GPU run 2: own complex Cholesky tests, two-context experiment with both factorisations
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dual.py tests/test_gpu_primal.py tests/test_gpu_estimator.py -m gpu -x -q > gpurun_out/r04/pytest_run2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_run2.log
tail -5 gpurun_out/r04/pytest_run2.log
for i in 1 2 3; do
  timeout 300 python tools/dev_two_context.py 125000 8 > gpurun_out/r04/two_ctx_own_$i.log 2>&1; echo "own $i rc $?"; tail -n 4 gpurun_out/r04/two_ctx_own_$i.log
  NLS_POTRF=rocsolver timeout 300 python tools/dev_two_context.py 125000 8 > gpurun_out/r04/two_ctx_rocsolver_$i.log 2>&1; echo "rocsolver $i rc $?"; tail -n 4 gpurun_out/r04/two_ctx_rocsolver_$i.log
done
timeout 300 python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_c2_b.json 2> gpurun_out/r04/bench_c2_b.err; echo "c2 rc $?"
timeout 300 python bench.py --config c3e --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/bench_c3e_b.json 2> gpurun_out/r04/bench_c3e_b.err; echo "c3e rc $?"
NLS_POTRF=rocsolver timeout 300 python bench.py --config c3e --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/bench_c3e_rocsolver.json 2> gpurun_out/r04/bench_c3e_rocsolver.err; echo "c3e rocsolver rc $?"
