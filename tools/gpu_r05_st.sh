# Round 5: per-kernel averages of a configuration (argument: c3e | c2 | c4), three fits.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/trST
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trST -- python3 bench.py --config $1 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1 ); echo "rc=$?"
python tools/kstats.py /tmp/trST k_trd k_zpotrf k_ztrsv k_dc k_sweep k_loo k_border k_gram_reduce k_featuremap Cijk | head -40
