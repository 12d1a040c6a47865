# Round-2 counter passes (one counter set per pass; kernel trace only, as gpurun requires).
#   rotate / gram: traffic past L2 and hit rate of k_rotate3 for the tile orders and K-walk phases under test
#   the same launches also hold k_featuremap / k_shift_pad rows (K1: FETCH_SIZE / WRITE_SIZE per launch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 0x0" "2 0x0" "4 0x0" "0 4x8" "2 4x8"; do
  set -- $cfg
  tag="ks$1_p$2"
  i=0
  for cset in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    NLS_ROT_KSTAGGER=$1 NLS_ROT_PATCH=$2 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcR2_${tag}_$i -- ./tools/nls_cbench 333440 128 4096 1024 rotate 1 > gpurun_out/pmcR2_${tag}_$i.log 2>&1; echo "$tag $i rc=$?"
  done
done
