"""kernels of the LAST step of a rocprofv3 --kernel-trace of the bench, grouped by (kernel, workgroups of the launch): launches, mean / min
duration, total per step -- what a family's small launches cost against its large ones.   usage: trace_by_grid.py <dir> [name filter]"""
import collections, csv, glob, re, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
rows = rows[adam[-2] + 1:adam[-1] + 1]


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", n)
    base = m.group(1) if m else n[:40]
    tmpl = (m.group(2) or "") if m else ""
    tmpl = re.sub(r"[A-Za-z_]+Cfg<([0-9, ]+).*", r"<\1>", tmpl)[:24]
    return base + tmpl


agg = collections.OrderedDict()
for r in rows:
    wg = 1
    for ax in "XYZ":
        wg *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
    name = short(r["Kernel_Name"])
    if flt and flt not in name:
        continue
    a = agg.setdefault((name, wg), [])
    a.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("%d kernels, %.2f ms of kernel time in the step" % (sum(len(v) for v in agg.values()), tot / 1e3))
fam = collections.defaultdict(float)
for (name, wg), v in agg.items():
    fam[name] += sum(v)
for name, t in sorted(fam.items(), key=lambda kv: -kv[1])[:40]:
    print("%8.3f ms  %s" % (t / 1e3, name))
    for (n2, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if n2 == name:
            print("              %6d workgroups x%-4d mean %8.1f us  min %8.1f  sum %7.3f ms" % (wg, len(v), sum(v) / len(v), min(v), sum(v) / 1e3))
