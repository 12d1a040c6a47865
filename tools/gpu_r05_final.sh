# Round 5, final evidence pass on the committed build: the whole GPU suite, smoke, the bench lines of every configuration, kernel stats of the
# bench command (rocprofv3 --kernel-trace --stats) for c3 / c2 / c3e / c4.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r05_smoke.log
python bench.py > gpurun_out/r05_bench_c3.json 2> gpurun_out/r05_bench_c3.err; echo "c3 rc=$?"
python bench.py --config c2 --steps 20 --warmup 3 > gpurun_out/r05_bench_c2.json 2> gpurun_out/r05_bench_c2.err; echo "c2 rc=$?"
python bench.py --config c3e --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r05_bench_c3e.json 2> gpurun_out/r05_bench_c3e.err; echo "c3e rc=$?"
python bench.py --config c4 --steps 10 --warmup 2 > gpurun_out/r05_bench_c4.json 2> gpurun_out/r05_bench_c4.err; echo "c4 rc=$?"
python bench.py --config c3i --steps 10 --warmup 2 --no-end-to-end > gpurun_out/r05_bench_c3i.json 2> gpurun_out/r05_bench_c3i.err; echo "c3i rc=$?"
for cfg in c3 c2 c3e c4; do
  steps=3; [ $cfg = c2 ] && steps=6; [ $cfg = c3e ] && steps=6; [ $cfg = c4 ] && steps=6
  rm -rf gpurun_out/prof_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$cfg -- python3 bench.py --config $cfg --steps $steps --warmup 0 --no-cpu-baseline --no-end-to-end > gpurun_out/r05_${cfg}_rocprof_bench_line.json 2> gpurun_out/r05_${cfg}_rocprof.err; echo "rocprof $cfg rc=$?"
  f=$(find gpurun_out/prof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_${cfg}_kernel_stats.csv
  rm -rf gpurun_out/prof_$cfg
done
ls -la gpurun_out | head -40
