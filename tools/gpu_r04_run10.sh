#!/bin/bash
mkdir -p gpurun_out/r04/prof_c4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dual.py tests/test_gpu_baseline_sizes.py tests/test_gpu_fullsize.py::test_c4_dual_identities tests/test_gpu_two_contexts.py -m gpu -x -q > gpurun_out/r04/pytest_run10.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_run10.log
tail -5 gpurun_out/r04/pytest_run10.log
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_c4 -o c4 -- python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/prof_c4/bench.json 2> gpurun_out/r04/prof_c4/bench.err
find gpurun_out/r04 -name "*kernel_trace.csv" -size +30M -delete
python - <<'PY'
import json,csv
for c in ("c4",):
    d=json.loads(open(f"gpurun_out/r04/prof_{c}/bench.json").read().strip().splitlines()[-1])
    print(c, d["value"], d["stage_ms_per_step"], d.get("evd_stage_ms"))
    rows=list(csv.DictReader(open(f"gpurun_out/r04/prof_{c}/{c}_kernel_stats.csv")))
    for r in rows[:40]:
        print("   %-70s calls %6d avg %9.1f us total/fit %8.2f ms"%(r["Name"][:70],int(r["Calls"]),float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6/4))
PY
