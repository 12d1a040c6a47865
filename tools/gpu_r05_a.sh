# Round 5, GPU pass A: the whole GPU suite on the group / grid build, the c2 idle probe, bench lines, counter passes, the full-size CPU Mode-S run.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r05a_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05a_pytest_gpu.log
python tools/probe_c2_idle.py > gpurun_out/r05_c2_idle.log 2>&1; echo "probe rc=$?"; cat gpurun_out/r05_c2_idle.log
python bench.py --config c2 --steps 20 --warmup 3 > gpurun_out/r05a_bench_c2.json 2> gpurun_out/r05a_bench_c2.err; echo "c2 rc=$?"
python bench.py > gpurun_out/r05a_bench_c3.json 2> gpurun_out/r05a_bench_c3.err; echo "c3 rc=$?"
python bench.py --config c3i --steps 10 --warmup 2 --no-end-to-end > gpurun_out/r05a_bench_c3i.json 2> gpurun_out/r05a_bench_c3i.err; echo "c3i rc=$?"
bash tools/pmc_passes_r05.sh > gpurun_out/r05a_pmc.log 2>&1; echo "pmc rc=$?"; tail -5 gpurun_out/r05a_pmc.log
python tools/cpu_modeS_full.py > gpurun_out/r05_cpu_modeS_c3_full.json 2> gpurun_out/r05_cpu_modeS_c3_full.err; echo "modeS rc=$?"
