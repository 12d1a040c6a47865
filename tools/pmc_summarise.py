"""Summarise the rocprofv3 --pmc passes of tools/evidence_pass.sh (section pmc): per kernel and configuration the fabric-side bytes
(2 x FETCH_SIZE as MI355X_MICROARCH.md prescribes for gfx950 wide reads, WRITE_SIZE as reported; both in KiB in the CSV),
the L2 hit rate and the kernel duration.  `--bench-layout <tag>`: the summary in the layout bench.py reads (profiles/<tag>_pmc_summary.json)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = defaultdict(lambda: defaultdict(dict))
for d in sorted(glob.glob(os.path.join(root, "pmc*_*_[0-9]"))):
    tag = re.match(r".*pmc(?:R2)?_(.*)_\d$", d).group(1)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("nls::", "").replace("void ", "")
            k = "k_featuremap" if k.startswith("k_featuremap") else k
            if not any(k.startswith(p) for p in ("k_rotate3", "k_featuremap", "k_shift_pad", "k_gram3", "k_sweep")):
                continue
            out[tag][k].setdefault(row["Counter_Name"], 0.0)
            out[tag][k][row["Counter_Name"]] += float(row["Counter_Value"])
            out[tag][k].setdefault("_disp_" + row["Counter_Name"], set()).add(row["Dispatch_Id"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("nls::", "").replace("void ", "")
            k = "k_featuremap" if k.startswith("k_featuremap") else k
            if k in out[tag]:
                out[tag][k].setdefault("_ns", []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
res = {}
for tag, ks in out.items():
    for k, c in ks.items():
        n = max(len(c.get("_disp_FETCH_SIZE", [])), 1)
        e = {"launches": n}
        if "FETCH_SIZE" in c: e["fetch_bytes_x2_per_launch"] = 2 * 1024 * c["FETCH_SIZE"] / n
        if "WRITE_SIZE" in c: e["write_bytes_per_launch"] = 1024 * c["WRITE_SIZE"] / max(len(c.get("_disp_WRITE_SIZE", [])), 1)
        if "TCC_HIT_sum" in c: e["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if "_ns" in c: e["avg_ms"] = sum(c["_ns"]) / len(c["_ns"]) / 1e6
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md): / 8 = chip cycles of the dispatch; 256 CUs x 4 SIMDs = 1024 matrix pipes
            nb = max(len(c.get("_disp_GRBM_GUI_ACTIVE", [])), 1)
            cyc = c["GRBM_GUI_ACTIVE"] / nb / 8.0
            e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / nb / (cyc * 1024.0)
            if "_ns" in c: e["clock_ghz"] = cyc / (sum(c["_ns"]) / len(c["_ns"]))
        res.setdefault(tag, {})[k] = e
if "--bench-layout" in sys.argv:
    rtag = sys.argv[sys.argv.index("--bench-layout") + 1]
    rows = 333440
    rot, fm, sp = res["rot_default"]["k_rotate3"], res["rot_default"]["k_featuremap"], res["rot_default"].get("k_shift_pad", {})

    def pick(e, **kw):
        o = {"rows_per_launch": rows, "D": 4096, "d": 128, "fetch_bytes_x2": e.get("fetch_bytes_x2_per_launch"), "write_bytes": e.get("write_bytes_per_launch"),
             "l2_hit": e.get("l2_hit"), "ms": e.get("avg_ms"), "mfma_busy": e.get("mfma_busy"), "clock_ghz": e.get("clock_ghz")}
        o.update(kw)
        return o

    res = {
        "_note": f"rocprofv3 --kernel-trace --pmc, one counter set per pass (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | GRBM_GUI_ACTIVE "
        f"SQ_VALU_MFMA_BUSY_CYCLES), tools/evidence_pass.sh {rtag} pmc; bytes past L2 = 2 x FETCH_SIZE (gfx950 wide reads, MI355X_MICROARCH.md) and "
        "WRITE_SIZE, KiB -> bytes; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 matrix pipes); 333 440 rows per launch, d = 128, D = 4096",
        "k_rotate3": pick(rot, order="default = 8-row XCD patch"),
        "k_gram3": pick(res["gram_default"]["k_gram3"], order="default on one GPU = plain"),
        "k_gram3_contiguous": pick(res["gram_contig"]["k_gram3"], order="XCD-contiguous (default with a communicator)"),
        "k_featuremap": {"D": 4096, "d": 128, "rows_per_launch": rows, "fetch_bytes_x2": fm.get("fetch_bytes_x2_per_launch"), "write_bytes": fm.get("write_bytes_per_launch"),
                         "ms": fm.get("avg_ms"),
                         "hbm_bytes_per_row": (fm.get("fetch_bytes_x2_per_launch", 0) + fm.get("write_bytes_per_launch", 0) + sp.get("fetch_bytes_x2_per_launch", 0)
                                               + sp.get("write_bytes_per_launch", 0)) / rows},
    }
print(json.dumps(res, indent=1))
