# Round 5, GPU pass L: kernel trace of a config-4 fit with the band look-ahead: do the chain and the update overlap?
# (NLS_SB_LOOKAHEAD exists only with profiles/r05_rejected/band_lookahead.diff.txt applied: the look-ahead was measured and removed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/trL
( timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trL -- python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r05l_trace_run.log 2>&1 ); echo "rc=$?"
python tools/trace_band_overlap.py /tmp/trL 2>&1 | tee gpurun_out/r05l_overlap.log
