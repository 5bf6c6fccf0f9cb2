"""Which kernels precede / follow a given kernel in a rocprofv3 --kernel-trace of the bench?  usage: trace_neighbors.py <dir> <substr>"""
import collections, csv, glob, sys
d, sub = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev, nxt = collections.Counter(), collections.Counter()
for i, r in enumerate(rows):
    if sub in r["Kernel_Name"]:
        if i: prev[rows[i - 1]["Kernel_Name"][:80]] += 1
        if i + 1 < len(rows): nxt[rows[i + 1]["Kernel_Name"][:80]] += 1
print("before:", prev.most_common(8))
print("after:", nxt.most_common(8))
