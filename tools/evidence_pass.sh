#!/bin/bash
# The ONE evidence script of a round (run through gpurun; everything lands under gpurun_out/<tag>/ and what is to be judged is copied into
# profiles/ by hand).  Usage:  bash tools/evidence_pass.sh <tag> <section> [<section> ...]
#   suite     the whole GPU test suite and smoke()
#   bench     the bench lines: the default command (c3), c2, c3e, c4, c3i, and one rank's share of an 8-GPU c3 fit (--as-rank 0 --of 8), one- and two-stage
#   stats     rocprofv3 --kernel-trace --stats of the bench command for c3 / c2 / c4 (kernel_stats.csv kept, traces deleted)
#   pmc       counter passes of the two MFMA kernels and the feature map (one counter set per pass, kernel trace only - gpurun refuses more;
#             the native driver tools/nls_cbench directly after "--"), summarised into <tag>_pmc_summary.json in the layout bench.py reads
#   evd       stage times of the eigendecompositions at the path's sizes, one- and two-stage (tools/time_evd.py, NLS_EVD_PROFILE=1)
#   predict   decision_function / predict_std rows per second (tools/time_predict.py)
# (Rounds 1-5 kept one script per GPU call, tools/gpu_r0x_*.sh, which the records under profiles/ of those rounds still name: they are in the
# history up to commit ccfcb54.)
tag=${1:?tag}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$tag; mkdir -p $O
for section in "$@"; do case $section in
suite)
  timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
bench)
  timeout 1500 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
  timeout 900 python bench.py --config c2 --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
  timeout 900 python bench.py --config c3e --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end > $O/bench_c3e.json 2> $O/bench_c3e.err; echo "c3e rc=$?"
  timeout 900 python bench.py --config c4 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
  timeout 900 python bench.py --config c3i --steps 10 --warmup 2 --no-end-to-end > $O/bench_c3i.json 2> $O/bench_c3i.err; echo "c3i rc=$?"
  for evd in onestage twostage; do
    NLS_EVD=$evd timeout 900 python bench.py --config c3 --as-rank 0 --of 8 --steps 10 --warmup 2 > $O/bench_c3_rank0of8_$evd.json 2> $O/bench_c3_rank0of8_$evd.err; echo "rank0of8 $evd rc=$?"
  done ;;
stats)
  for cfg in c3 c2 c4; do
    steps=3; [ $cfg != c3 ] && steps=6
    rm -rf $O/prof_$cfg
    timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python3 bench.py --config $cfg --steps $steps --warmup 0 --no-cpu-baseline --no-end-to-end > $O/${cfg}_rocprof_bench_line.json 2> $O/${cfg}_rocprof.err; echo "rocprof $cfg rc=$?"
    f=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${cfg}_kernel_stats.csv
    rm -rf $O/prof_$cfg
  done ;;
pmc)
  g++ -O2 tools/nls_cbench.cpp -Iinclude -Lneo_ls_svm_amd -lneolssvm_hip -Wl,-rpath,$PWD/neo_ls_svm_amd -o /tmp/nls_cbench || exit 1
  /tmp/nls_cbench 8192 128 4096 1024 rotate 1 > /dev/null 2>&1   # page the libraries in before the first profiled pass
  run() {  # tag, what, env...
    t=$1; what=$2; shift 2; i=0
    for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
      i=$((i+1))
      ( export "$@"; timeout 300 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d $O/pmc_${t}_$i -- /tmp/nls_cbench 333440 128 4096 1024 $what 1 > $O/pmc_${t}_$i.log 2>&1 ); echo "pmc $t $i rc=$?"
    done
  }
  run rot_default rotate NLS_DUMMY=1
  run gram_default gram NLS_DUMMY=1
  run gram_contig gram NLS_GRAM_ORDER=contiguous
  python tools/pmc_summarise.py $O --bench-layout $tag > $O/${tag}_pmc_summary.json; tail -c 600 $O/${tag}_pmc_summary.json
  find $O -path "*pmc_*" -name "*.csv" -size +2M -delete ;;
evd)
  export NLS_EVD_PROFILE=1
  for cfg in "1025 c" "4097 c" "6000 r" "10000 r"; do set -- $cfg
    for mode in onestage twostage; do echo "== n=$1 $2 $mode"; NLS_EVD=$mode timeout 600 python tools/time_evd.py $1 $2 3 2>&1 | grep -v "n=64" | tail -4; done
  done > $O/evd_stages.log 2>&1
  unset NLS_EVD_PROFILE; grep "two-stage\|one-stage" $O/evd_stages.log | tail -12 ;;
predict)
  timeout 900 python tools/time_predict.py > $O/time_predict.log 2>&1; echo "predict rc=$?"; tail -8 $O/time_predict.log ;;
*) echo "unknown section $section" ;;
esac; done
