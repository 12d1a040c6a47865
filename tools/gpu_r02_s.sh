#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for what in rotate gram; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/pmcS_${what} -- ./tools/nls_cbench 333440 128 4096 1024 $what 1 > gpurun_out/pmcS_${what}.log 2>&1; echo "$what rc=$?"
done
python - <<'PY'
import csv, glob
for what in ("rotate","gram"):
    acc={}
    for f in glob.glob(f"gpurun_out/pmcS_{what}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if "k_rotate3" in k or "k_gram3" in k:
                acc.setdefault(k,{}).setdefault(r["Counter_Name"],0.0)
                acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    dur={}
    for f in glob.glob(f"gpurun_out/pmcS_{what}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if k in acc: dur.setdefault(k,[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k,c in acc.items():
        t=sum(dur[k])/1e9
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs of the chip (1024); GRBM_GUI_ACTIVE counts per-XCD cycles summed over 8 XCDs
        clk=c["GRBM_GUI_ACTIVE"]/8/t
        print(k, "duration %.1f ms"%(t*1e3), "clock %.3f GHz"%(clk/1e9), "MFMA busy %.3f"%(c["SQ_VALU_MFMA_BUSY_CYCLES"]/(c["GRBM_GUI_ACTIVE"]/8*1024)), c)
PY
