# Round 5, GPU pass K: compute-unit masks (probe), then the band reduction's look-ahead on masked streams: two-stage tests, config 4 with
# (NLS_SB_LOOKAHEAD exists only with profiles/r05_rejected/band_lookahead.diff.txt applied: the look-ahead was measured and removed)
# the look-ahead off / on masked streams / on plain streams / other side-set sizes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twostage.py -m gpu -x -q 2>&1 | tail -3
run() {
  timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 > gpurun_out/r05k_c4_$1.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05k_c4_$1.json").read())
print("$1", round(d["ms_per_step"],1), d.get("evd_stage_ms"))
PY
}
NLS_SB_LOOKAHEAD=0 run off
run masked8
NLS_SB_LOOKAHEAD=plain run plain
NLS_SB_SIDE_EVERY=4 run masked4
NLS_SB_SIDE_EVERY=16 run masked16
NLS_SB_LOOKAHEAD=0 run off2
run masked8_2
