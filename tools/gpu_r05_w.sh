# Round 5, GPU pass W: default back-transformation (V^H explicit for complex): primal / EVD tests, configs 2 / 3e with the knob off and on.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_evd.py tests/test_gpu_primal.py tests/test_gpu_twostage.py -m gpu -x -q 2>&1 | tail -2
for v in 0 1 0 1; do
for c in c2 c3e; do
NLS_BT_VT=$v timeout 300 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05w_$c.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05w_$c.json").read())
print("vt=$v $c", round(d["ms_per_step"],2), d.get("evd_stage_ms"))
PY
done
done
