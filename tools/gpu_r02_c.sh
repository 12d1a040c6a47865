#!/bin/bash
# K1 stagger sweep, rotate K-walk stagger / patch sweep (timing through the native driver), failed tests again
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -k "ames or badly_scaled" > gpurun_out/r02c_pytest.log 2>&1; tail -5 gpurun_out/r02c_pytest.log
for us in 0 10 20 30 40 60; do
  echo "== K1 stagger $us us"; NLS_K1_STAGGER_US=$us ./tools/nls_cbench 1000000 128 4096 32 fit 3 2>&1 | tail -2
done > gpurun_out/r02c_k1_stagger.log 2>&1
cat gpurun_out/r02c_k1_stagger.log
for cfg in "0 0x0" "2 0x0" "3 0x0" "0 4x8" "2 4x8" "3 4x8" "4 4x8" "2 2x16" "2 8x4"; do
  set -- $cfg
  echo "== rotate kstagger $1 patch $2"; NLS_ROT_KSTAGGER=$1 NLS_ROT_PATCH=$2 ./tools/nls_cbench 333440 128 4096 1024 rotate 3 2>&1 | tail -2
done > gpurun_out/r02c_rot_stagger.log 2>&1
cat gpurun_out/r02c_rot_stagger.log
