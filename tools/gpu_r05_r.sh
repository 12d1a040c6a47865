# Round 5, GPU pass R: DPP sums in the one-stage tridiagonalisation's row kernels: EVD tests, configs 2 / 3e / 4.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_twostage.py tests/test_gpu_evd.py tests/test_gpu_stedc.py tests/test_gpu_primal.py -m gpu -x -q 2>&1 | tail -3
for c in c2 c3e c4; do
timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05r_$c.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05r_$c.json").read())
print("$c", round(d["ms_per_step"],2), d["value"], d.get("evd_stage_ms"))
PY
done
