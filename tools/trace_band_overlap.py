"""Reads a rocprofv3 kernel trace (CSV) of a config-4 fit and prints the time line of a few panels of the band reduction: which kernels of the
factorisation chain ran beside the trailing update (k_sb_her2k), on which hardware queue."""
import csv, glob, sys
path = [p for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)][0]
rows = list(csv.DictReader(open(path)))
def short(n):
    n = n.split("(")[0]
    return n.split("::")[-1][:28]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows]
ev.sort()
sb = [e for e in ev if e[2].startswith("k_sb_")]
print("kernels", len(ev), "band-reduction kernels", len(sb), "queues", sorted({e[3] for e in sb}))
# the last fit's band reduction: take the last 14 * 312 kernels, show panels 100..102
her = [i for i, e in enumerate(sb) if e[2].startswith("k_sb_hemm<")]
if len(her) > 420:
    i0 = her[-212]
    t0 = sb[i0][0]
    for e in sb[i0:i0 + 48]:
        print(f"{(e[0]-t0)/1e3:9.1f} us  +{(e[1]-e[0])/1e3:7.1f}  q{e[3]}  grid {e[4]:8d}  {e[2]}")
# overlap: time during which a her2k and a chain kernel are both running
chain = ("k_sb_gram", "k_sb_small", "k_sb_apply", "k_sb_finish")
h = [(e[0], e[1]) for e in sb if e[2].startswith("k_sb_her2k")]
c = [(e[0], e[1]) for e in sb if e[2].startswith(chain)]
tot = 0
j = 0
for a0, a1 in h:
    while j < len(c) and c[j][1] < a0: j += 1
    k = j
    while k < len(c) and c[k][0] < a1:
        tot += max(0, min(a1, c[k][1]) - max(a0, c[k][0])); k += 1
print("her2k time", sum(b - a for a, b in h) / 1e6, "ms; chain time", sum(b - a for a, b in c) / 1e6, "ms; both running", tot / 1e6, "ms")
