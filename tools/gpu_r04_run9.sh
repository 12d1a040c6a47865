#!/bin/bash
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_run9.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/pytest_run9.log
tail -6 gpurun_out/r04/pytest_run9.log
