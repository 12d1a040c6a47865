#!/bin/bash
O=gpurun_out/r04/prof_c4b
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c4 -- python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench.json 2> $O/bench.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r04/prof_c4b/c4_kernel_stats.csv")))
for r in rows[:45]:
    n=r["Name"]
    print(f"{n[:64]:64s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/4e6:8.2f} ms/fit avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
