# Round 5, GPU pass N: per-kernel averages of the band reduction after the 64-row panel kernels (config 4, three fits).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/trN
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trN -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1 ); echo "rc=$?"
python tools/kstats.py /tmp/trN k_sb_ k_potrf k_chase k_q2 | tee gpurun_out/r05n_sb_stats.log
