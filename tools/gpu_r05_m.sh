# Round 5, GPU pass M: panel kernels of the band reduction with 64-row blocks / four threads per row: tests, config 4 (look-ahead off / on),
# (NLS_SB_LOOKAHEAD exists only with profiles/r05_rejected/band_lookahead.diff.txt applied: the look-ahead was measured and removed)
# kernel stats of one run.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twostage.py -m gpu -x -q 2>&1 | tail -3
run() {
  timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05m_c4_$1.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05m_c4_$1.json").read())
print("$1", round(d["ms_per_step"],1), d.get("evd_stage_ms"))
PY
}
NLS_SB_LOOKAHEAD=0 run off
run masked8
NLS_SB_LOOKAHEAD=plain run plain
NLS_SB_LOOKAHEAD=0 run off2
run masked8_2
rm -rf /tmp/trM
( NLS_SB_LOOKAHEAD=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trM -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1 ); echo "rc=$?"
f=$(find /tmp/trM -name "*kernel_stats.csv" | head -1); grep -E "k_sb_|Name" $f | cut -d, -f1-4 | cut -c1-150
