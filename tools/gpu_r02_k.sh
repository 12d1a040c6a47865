#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02k_prof -- python3 tools/time_dual.py > gpurun_out/r02k_dual.log 2>&1
find gpurun_out/r02k_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02k_dual_kernel_stats.csv; rm -rf gpurun_out/r02k_prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02k_dual_kernel_stats.csv')))
for r in rows[:22]:
    print(f"{r['Name'].split('(')[0][-60:]:60s} calls {r['Calls']:>7s} total {float(r['TotalDurationNs'])/1e6:9.1f} ms avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
