# Round 5, GPU pass Y: does routing rocBLAS's GEMMs through hipBLASLt change the first back-transformation (config 4: Q1)?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 0 1; do
ROCBLAS_USE_HIPBLASLT=$v timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05y_c4.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05y_c4.json").read())
print("hipblaslt=$v c4", round(d["ms_per_step"],2), d.get("evd_stage_ms"))
PY
done
