# Round 5, GPU pass Z: hipBLASLt behind rocBLAS for the other configurations (complex back-transformation, small sizes) and the first-call cost.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
for c in c3e c2; do
ROCBLAS_USE_HIPBLASLT=$v timeout 300 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05z_$c.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05z_$c.json").read())
print("hipblaslt=$v $c", round(d["ms_per_step"],2), d.get("evd_stage_ms"))
PY
done
/usr/bin/time -f "hipblaslt=$v smoke wall %e s" env ROCBLAS_USE_HIPBLASLT=$v python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
done
ROCBLAS_USE_HIPBLASLT=1 timeout 1500 python -m pytest tests/test_gpu_evd.py tests/test_gpu_twostage.py tests/test_gpu_dual.py tests/test_gpu_two_contexts.py -m gpu -x -q 2>&1 | tail -2
