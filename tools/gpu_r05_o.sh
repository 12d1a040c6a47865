# Round 5, GPU pass O: one-wave small factorisations / series / quad solves in the band reduction's small kernels: tests, time line, config 4, stats.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twostage.py tests/test_gpu_evd.py -m gpu -x -q 2>&1 | tail -3
NLS_SB_STAMP=1 timeout 300 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end 2>&1 | grep -E "time line" | head -2
for i in 1 2; do
timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05o_c4_$i.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05o_c4_$i.json").read())
print(round(d["ms_per_step"],1), d.get("evd_stage_ms"), d["parity"] if "parity" in d else "")
PY
done
rm -rf /tmp/trO
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trO -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1 ); echo "rc=$?"
python tools/kstats.py /tmp/trO k_sb_ | tee gpurun_out/r05o_sb_stats.log
