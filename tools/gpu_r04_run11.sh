#!/bin/bash
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes_r04.sh > gpurun_out/r04/pmc_passes.log 2>&1
tail -20 gpurun_out/r04/pmc_passes.log | cut -c1-300
