# Round 5, GPU pass U: K1 epilogue (row scales fetched before the stores; DPP sums in the fused gemv): primal / estimator tests, configs 3e / 2 / 3.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_primal.py tests/test_gpu_estimator.py tests/test_gpu_twostage.py -m gpu -x -q 2>&1 | tail -3
for c in c3e c2; do
timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05u_$c.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05u_$c.json").read())
print("$c", round(d["ms_per_step"],2), d["value"], {k:v for k,v in d["stage_ms_per_step"].items() if k in ("featuremap","cholesky","evd","gram","rotate")})
PY
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05u_c3.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05u_c3.json").read())
print("c3", round(d["ms_per_step"],2), d["value"], d["stage_ms_per_step"], d.get("roofline_k1"))
PY
