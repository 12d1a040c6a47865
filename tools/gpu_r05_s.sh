# Round 5, GPU pass S: batched loads at the head of the small kernels (block_copy), Cholesky panel prologue, triangular solves: tests, configs 4 / 2 / 3e, stats of c4.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_twostage.py tests/test_gpu_evd.py tests/test_gpu_dual.py tests/test_gpu_primal.py tests/test_gpu_baseline_sizes.py -m gpu -x -q 2>&1 | tail -3
for c in c4 c2 c3e; do
timeout 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05s_$c.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05s_$c.json").read())
print("$c", round(d["ms_per_step"],2), d["value"], d.get("evd_stage_ms"), {k:v for k,v in d["stage_ms_per_step"].items() if k in ("cholesky","evd","gram","rotate")})
PY
done
rm -rf /tmp/trS
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trS -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1 ); echo "rc=$?"
python tools/kstats.py /tmp/trS k_sb_ k_potrf k_trsv | tee gpurun_out/r05s_stats.log
