# Round 5, GPU pass V: first back-transformation with V^H stored explicitly (NLS_BT_VT=1: no-transpose GEMMs) against the default.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NLS_BT_VT=1 timeout 900 python -m pytest tests/test_gpu_twostage.py tests/test_gpu_evd.py -m gpu -x -q 2>&1 | tail -2
for v in 0 1 0 1; do
for c in c4 c3e; do
NLS_BT_VT=$v timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05v_$c.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05v_$c.json").read())
print("vt=$v $c", round(d["ms_per_step"],2), d.get("evd_stage_ms"))
PY
done
done
