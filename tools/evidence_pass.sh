#!/bin/bash
# Evidence pass of a round (run through gpurun; results land in gpurun_out/r03final and are copied into profiles/ by hand): build check + smoke, full GPU tests, bench lines of every config, kernel stats of the c3 and c4 lines, numerics, EVD stage times
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r03final; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?" > $O/summary.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc $?" >> $O/summary.txt
timeout 1200 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc $?" >> $O/summary.txt
for c in c2 c3e c4 c5; do timeout 1500 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc $?" >> $O/summary.txt; done
NLS_EVD=onestage timeout 900 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_onestage.json 2>/dev/null
timeout 600 python tools/numerics_report.py > $O/numerics.txt 2>&1; echo "numerics rc $?" >> $O/summary.txt
export NLS_EVD_PROFILE=1
for cfg in "1025 c" "4097 c" "6000 r" "8000 r" "10000 r"; do set -- $cfg
  for mode in onestage twostage; do echo "== n=$1 $2 $mode"; NLS_EVD=$mode timeout 600 python tools/time_evd.py $1 $2 3 2>&1 | grep -v "n=64" | tail -4; done
done > $O/evd_stages.log 2>&1
unset NLS_EVD_PROFILE
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_c3" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$O/prof_c3_line.json" 2> "$GRAFT_REPO_ROOT/$O/prof_c3.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof_c4" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config c4 --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$O/prof_c4_line.json" 2> "$GRAFT_REPO_ROOT/$O/prof_c4.err"
cd "$GRAFT_REPO_ROOT"; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/summary.txt; tail -n 3 $O/pytest_gpu.log; grep "two-stage" $O/evd_stages.log | tail -8
