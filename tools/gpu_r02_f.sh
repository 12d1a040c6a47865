#!/bin/bash
mkdir -p gpurun_out
for b in 0 1; do echo "== boustrophedon $b"; NLS_TRD_BOUSTROPHEDON=$b python tools/time_evd.py 10000 r 3; NLS_TRD_BOUSTROPHEDON=$b python tools/time_evd.py 6500 r 2; NLS_TRD_BOUSTROPHEDON=$b python tools/time_evd.py 4097 c 2; done > gpurun_out/r02f_evd.log 2>&1
cat gpurun_out/r02f_evd.log
python tools/time_dual.py > gpurun_out/r02f_dual_c4.log 2>&1; tail -4 gpurun_out/r02f_dual_c4.log
python -m pytest tests/test_gpu_evd.py tests/test_gpu_baseline_sizes.py -m gpu -q -k "eigh or trid" 2>&1 | tail -3
