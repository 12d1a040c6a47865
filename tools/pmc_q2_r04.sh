# Counter passes for the second back-transformation (real n = 10^4): which pipe is busy.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
( timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcQ2_warm -- python3 tools/time_evd.py 1000 r 1 > gpurun_out/r04/pmcQ2_warm.log 2>&1 ); echo "warm rc=$?"
i=0
for cset in "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for form in wave team; do
    ( export NLS_Q2_FORM=$form; timeout 200 rocprofv3 --kernel-trace --pmc $cset --output-format csv -d gpurun_out/pmcQ2_${form}_$i -- python3 tools/time_evd.py 10000 r 1 > gpurun_out/r04/pmcQ2_${form}_$i.log 2>&1 ); echo "$form $i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, collections
for form in ("wave","team"):
    tot=collections.defaultdict(float)
    for f in glob.glob(f"gpurun_out/pmcQ2_{form}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_q2_apply" in r["Kernel_Name"]:
                tot[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(form, dict(tot))
PY
find gpurun_out -path "*pmcQ2_*" -name "*.csv" -size +2M -delete
