#!/bin/bash
# Round 4, second final call (after the dual output-path work): whole GPU suite, smoke, c4 / c3 / c3e / c2 lines, c4 kernel statistics.
O=gpurun_out/r04/final2
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 600 python bench.py --config c4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc $?"
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc $?"
timeout 400 python bench.py --config c3e --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_c3e.json 2> $O/bench_c3e.err; echo "c3e rc $?"
timeout 400 python bench.py --config c2 --steps 10 --warmup 2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc $?"
mkdir -p $O/prof_c4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/prof_c4/bench.json 2> $O/prof_c4/bench.err; echo "prof c4 rc $?"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/final2/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], "e2e", d.get("value_end_to_end"), "roof", (d.get("roofline") or {}).get("frac"), "parity", (d.get("parity") or {}).get("loo_residuals_max_rel_err"))
    except Exception as e: print(f, "ERR", e)
PY
