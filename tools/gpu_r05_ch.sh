# Round 5: Cholesky-related tests and the Cholesky stage of configs 3e / 2.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_primal.py tests/test_gpu_dual.py -m gpu -x -q 2>&1 | tail -2
for c in c3e c2 c3e c2; do
timeout 300 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > /tmp/ch.json
python - <<PY
import json
d=json.loads(open("/tmp/ch.json").read())
print("$c", round(d["ms_per_step"],2), "cholesky", d["stage_ms_per_step"]["cholesky"], "evd", d["stage_ms_per_step"]["evd"])
PY
done
