"""Prints the kernels of a rocprofv3 --stats run (CSV) whose names contain one of the given substrings: calls, average (us), total (ms)."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
pats = sys.argv[2:] or [""]
for r in csv.DictReader(open(path)):
    if any(p in r["Name"] for p in pats):
        print(f'{r["Name"].split("(")[0][-46:]:46s} calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:8.1f} us  total {int(r["TotalDurationNs"])/1e6:8.2f} ms')
