#!/bin/bash
# Round 4, last measurement call: whole GPU suite, smoke, every bench line (warm-up 2: the pooled output buffers exist before the timed steps),
# kernel statistics of c3e / c2 / c4.
O=gpurun_out/r04/final3
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 3 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc $?"
timeout 400 python bench.py --config c2 --steps 10 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc $?"
timeout 400 python bench.py --config c3e --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/bench_c3e.json 2> $O/bench_c3e.err; echo "c3e rc $?"
timeout 600 python bench.py --config c4 --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc $?"
timeout 900 python bench.py --config c5 --steps 1 --warmup 0 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc $?"
for cfg in c3e c2 c4; do
  mkdir -p $O/prof_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -o $cfg -- python3 bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/prof_$cfg/bench.json 2> $O/prof_$cfg/bench.err; echo "prof $cfg rc $?"
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/final3/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], "e2e", d.get("value_end_to_end"), "roof", (d.get("roofline") or {}).get("frac"), "parity", (d.get("parity") or {}).get("loo_residuals_max_rel_err"))
    except Exception as e: print(f, "ERR", e)
PY
