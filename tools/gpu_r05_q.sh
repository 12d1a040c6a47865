# Round 5, GPU pass Q: read-modify-write epilogues loaded in batches (k_potrf_syrk, k_zpotrf_herk, k_sb_her2k, k_trd_rank2k, k_sb_x): tests, configs 4 / 2 / 3e.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_twostage.py tests/test_gpu_evd.py tests/test_gpu_dual.py tests/test_gpu_primal.py -m gpu -x -q 2>&1 | tail -3
for c in c4 c2 c3e; do
timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05q_$c.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05q_$c.json").read())
print("$c", round(d["ms_per_step"],2), d["value"], d.get("evd_stage_ms"), {k:v for k,v in d["stage_ms_per_step"].items() if k in ("cholesky","evd","gram","rotate")})
PY
done
