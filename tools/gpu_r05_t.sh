# Round 5, GPU pass T: quick check of a band-reduction change: two-stage tests, config 4 twice, band kernel stats.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twostage.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05t_c4_$i.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r05t_c4_$i.json").read())
print(round(d["ms_per_step"],1), d.get("evd_stage_ms"))
PY
done
rm -rf /tmp/trT
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trT -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1 ); echo "rc=$?"
python tools/kstats.py /tmp/trT k_sb_hemm k_sb_her2k | tee gpurun_out/r05t_stats.log
