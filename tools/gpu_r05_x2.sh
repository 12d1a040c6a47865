# Round 5, GPU pass X2: config 2 with rocBLAS's GEMMs through hipBLASLt or not, alternating, 30 steps each (box noise).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 0 1 0 1; do
ROCBLAS_USE_HIPBLASLT=$v timeout 300 python bench.py --config c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05x2_c2.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05x2_c2.json").read())
print("hipblaslt=$v c2", round(d["ms_per_step"],2), d["stage_ms_per_step"]["evd"], d["stage_ms_per_step"]["cholesky"], "e2e", d.get("value_end_to_end"))
PY
done
