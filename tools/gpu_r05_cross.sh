# Round 5, GPU pass: crossover of the real one-stage / two-stage eigendecomposition after the band reduction's round-5 work (device time via NLS_EVD_PROFILE).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 2048 3000 4000 5000 6000; do
  for mode in onestage twostage; do
    NLS_EVD=$mode NLS_EVD_PROFILE=1 timeout 300 python tools/time_evd.py $n r 3 2>&1 | grep -E "total|eigh n" | tail -2 | tr '\n' ' '; echo " [$mode]"
  done
done
