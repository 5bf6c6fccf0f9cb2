"""The kernels of the LAST step of a rocprofv3 --kernel-trace of the bench in start order: offset from the step's first kernel (ms), duration
(us), queue, short name -- to read the serial sections (heads, loss) launch by launch.   usage: trace_sequence.py <dir> [min_us]"""
import csv, glob, re, sys
d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
ev = ev[adam[-2] + 1:adam[-1] + 1]


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", n)
    base = m.group(1) if m else n[:40]
    tmpl = (m.group(2) or "") if m else ""
    tmpl = re.sub(r"[A-Za-z_]+Cfg<([0-9, ]+).*", r"<\1>", tmpl)[:30]
    return base + tmpl


t0 = ev[0][0]
qs = {}
prev_end = t0
for s, e, n, q in ev:
    qi = qs.setdefault(q, len(qs))
    if (e - s) / 1e3 >= min_us:
        print("%9.3f %8.1f q%d gap%7.1f %s" % ((s - t0) / 1e6, (e - s) / 1e3, qi, (s - prev_end) / 1e3, short(n)))
    prev_end = max(prev_end, e)
print("step wall %.3f ms, %d kernels" % ((ev[-1][1] - t0) / 1e6, len(ev)))
