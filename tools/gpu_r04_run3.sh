#!/bin/bash
# round 4, GPU run 3: fused complex Cholesky panel kernel, two-context tests, rocSOLVER concurrency probe, end-to-end profiles
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dual.py tests/test_gpu_two_contexts.py -m gpu -x -q > gpurun_out/r04/pytest_run3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_run3.log
tail -5 gpurun_out/r04/pytest_run3.log
/opt/rocm/bin/hipcc -O2 tools/probe_rocsolver_concurrency.cpp -o /tmp/probe_rsc -lrocsolver -lrocblas 2> gpurun_out/r04/probe_build.err && (timeout 300 /tmp/probe_rsc 512 40; timeout 300 /tmp/probe_rsc 1024 20) > gpurun_out/r04/probe_rocsolver_concurrency.log 2>&1
cat gpurun_out/r04/probe_rocsolver_concurrency.log
timeout 300 python bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_c2_c.json 2> gpurun_out/r04/bench_c2_c.err; echo "c2 rc $?"
timeout 300 python bench.py --config c3e --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/bench_c3e_c.json 2> gpurun_out/r04/bench_c3e_c.err; echo "c3e rc $?"
timeout 300 python tools/profile_fit.py 100000 64 1024 > gpurun_out/r04/profile_fit_c2.log 2>&1; echo "profile c2 rc $?"
timeout 600 python tools/profile_fit.py 1000000 128 4096 > gpurun_out/r04/profile_fit_c3.log 2>&1; echo "profile c3 rc $?"
python - <<'PY'
import json
for c in ("c2_c","c3e_c"):
    d=json.loads(open(f"gpurun_out/r04/bench_{c}.json").read().strip().splitlines()[-1])
    print(c, d["value"], d["stage_ms_per_step"], d.get("end_to_end"))
PY
