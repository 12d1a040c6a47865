#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02final_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02final_pytest.log; tail -5 gpurun_out/r02final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r02final_bench.json 2> gpurun_out/r02final_bench.err; echo "bench rc=$?"; wc -l gpurun_out/r02final_bench.json; tail -c 1500 gpurun_out/r02final_bench.json
python bench.py --config c2 --steps 10 --warmup 2 > gpurun_out/r02final_bench_c2.json 2>/dev/null; tail -c 900 gpurun_out/r02final_bench_c2.json
