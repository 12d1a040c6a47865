# Round 5, GPU pass X: config 2 / 3 with the back-transformation's V^H knob off and on, more steps (box noise).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1 0 1 0 1; do
NLS_BT_VT=$v timeout 300 python bench.py --config c2 --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05x_c2.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05x_c2.json").read())
print("vt=$v c2", round(d["ms_per_step"],2), d["stage_ms_per_step"]["evd"], d["stage_ms_per_step"]["total"], d.get("evd_stage_ms"))
PY
done
