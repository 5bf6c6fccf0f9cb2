"""kernels of the last step of a rocprofv3 kernel trace whose launch has FEWER workgroups than the chip has CUs (256), by total time: the
one-block-chain reductions of the OCR head were found this way"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
rows = rows[adam[-2] + 1:adam[-1] + 1]
agg = collections.OrderedDict()
for r in rows:
    wg = 1
    for d in "XYZ":
        wg *= max(1, int(r["Grid_Size_" + d]) // max(1, int(r["Workgroup_Size_" + d])))
    if wg >= 256:
        continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    a = agg.setdefault((name, wg), [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values())
print("launches with < 256 workgroups: %.2f ms of kernel time per step" % tot)
for (name, wg), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%8.3f ms  x%-4d  %4d workgroups  %s" % (ms, n, wg, name))
