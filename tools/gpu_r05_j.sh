# Round 5, GPU pass J: look-ahead of the real Cholesky factorisation (dual path): parity tests, then config 4 with and without it.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dual.py tests/test_gpu_two_contexts.py tests/test_gpu_baseline_sizes.py -m gpu -x -q 2>&1 | tail -3
for la in 1 0 1 0; do
  NLS_POTRF_LOOKAHEAD=$la timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05j_c4_la$la.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05j_c4_la$la.json").read())
print("lookahead=$la", d["ms_per_step"], d.get("stages"))
PY
done
