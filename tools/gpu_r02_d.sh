#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r02d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02d_pytest.log
tail -25 gpurun_out/r02d_pytest.log
python tools/time_predict.py > gpurun_out/r02d_time_predict.log 2>&1; tail -4 gpurun_out/r02d_time_predict.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02d_bench_c3.json 2> gpurun_out/r02d_bench_c3.err; echo "bench rc=$?"; tail -c 900 gpurun_out/r02d_bench_c3.json
bash tools/pmc_passes.sh
python tools/pmc_summarise.py gpurun_out > gpurun_out/r02d_pmc_summary.json 2> gpurun_out/r02d_pmc_summary.err; cat gpurun_out/r02d_pmc_summary.json | head -120; tail -5 gpurun_out/r02d_pmc_summary.err
