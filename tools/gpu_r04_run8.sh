#!/bin/bash
mkdir -p gpurun_out/r04/prof_c3e gpurun_out/r04/prof_c2 gpurun_out/r04/prof_c4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stedc.py tests/test_gpu_dual.py tests/test_gpu_two_contexts.py tests/test_gpu_primal.py tests/test_gpu_evd.py tests/test_gpu_twostage.py -m gpu -x -q > gpurun_out/r04/pytest_run8.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_run8.log
tail -6 gpurun_out/r04/pytest_run8.log
for c in c2 c3e c4; do
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_$c -o $c -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/prof_$c/bench.json 2> gpurun_out/r04/prof_$c/bench.err
done
find gpurun_out/r04 -name "*kernel_trace.csv" -size +30M -delete
python - <<'PY'
import json,csv
for c in ("c2","c3e","c4"):
    try:
        d=json.loads(open(f"gpurun_out/r04/prof_{c}/bench.json").read().strip().splitlines()[-1])
        print(c, d["value"], d["stage_ms_per_step"], d.get("evd_stage_ms"))
        rows=list(csv.DictReader(open(f"gpurun_out/r04/prof_{c}/{c}_kernel_stats.csv")))
        for r in rows:
            if any(k in r["Name"] for k in ("zpotrf","ztrsv","k_dc_","potrf","trsv")):
                print("   * %-68s calls %6d avg %9.1f us total %8.2f ms"%(r["Name"][:68],int(r["Calls"]),float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6))
    except Exception as e:
        print(c,"ERR",e)
PY
