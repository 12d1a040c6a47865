#!/bin/bash
mkdir -p gpurun_out/r04/prof_c3e gpurun_out/r04/prof_c2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_c3e -o c3e -- python3 bench.py --config c3e --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/prof_c3e/bench.json 2> gpurun_out/r04/prof_c3e/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_c2 -o c2 -- python3 bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r04/prof_c2/bench.json 2> gpurun_out/r04/prof_c2/bench.err
find gpurun_out/r04/prof_c3e gpurun_out/r04/prof_c2 -name "*kernel_stats*" | head
for f in $(find gpurun_out/r04/prof_c3e gpurun_out/r04/prof_c2 -name "*kernel_stats.csv"); do echo $f; head -22 $f | cut -c1-160; done
find gpurun_out/r04 -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/r04 -name "*.db" -delete
