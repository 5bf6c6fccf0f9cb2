"""kernel counts and summed durations per (queue, stream) of a rocprofv3 kernel trace over the last step (delimited by adam_kernel)"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
rows = rows[adam[-2] + 1:adam[-1] + 1]
keys = [k for k in ("Queue_Id", "Stream_Id", "Thread_Id") if k in rows[0]]
print("columns:", list(rows[0].keys()))
agg = collections.OrderedDict()
for r in rows:
    k = tuple(r[c] for c in keys)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, (n, ms) in agg.items():
    print(dict(zip(keys, k)), "kernels %d  summed %.2f ms" % (n, ms))
