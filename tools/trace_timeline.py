"""Where the GPU runs ONE kernel at a time: from a rocprofv3 --kernel-trace of the bench (steps delimited by adam_kernel) the concurrency
profile of the last steps (time with 0 / 1 / 2 / 3 / >= 4 kernels in flight) and the kernels ranked by EXCLUSIVE time (time during which
a launch is the only kernel on the chip: the serial path of the step), next to their total time.   usage: trace_timeline.py <dir> [steps]"""
import collections, csv, glob, re, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
lo, hi = adam[-steps - 1] + 1, adam[-1] + 1
ev = ev[lo:hi]


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*>)?", n)
    base = m.group(1) if m else n[:40]
    tmpl = (m.group(2) or "") if m else ""
    tmpl = re.sub(r"[A-Za-z_]+Cfg<([0-9, ]+).*", r"<\1>", tmpl)[:28]
    return base + tmpl


pts = []
for i, (s, e, n) in enumerate(ev):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort()
active = set()
conc = collections.Counter()
excl = collections.Counter()
tot = collections.Counter()
cnt = collections.Counter()
prev = pts[0][0]
for t, k, i in pts:
    dt = t - prev
    if dt > 0:
        conc[min(len(active), 4)] += dt
        if len(active) == 1:
            excl[short(ev[next(iter(active))][2])] += dt
    prev = t
    if k == 1:
        active.add(i)
    else:
        active.discard(i)
for s, e, n in ev:
    tot[short(n)] += e - s
    cnt[short(n)] += 1
wall = pts[-1][0] - pts[0][0]
print("steps %d: wall %.2f ms/step, %d kernels/step" % (steps, wall / steps / 1e6, len(ev) // steps))
print("kernels in flight (ms/step):  none %.2f   one %.2f   two %.2f   three %.2f   four or more %.2f" % tuple(conc[k] / steps / 1e6 for k in range(5)))
print("%-58s %9s %9s %8s %9s" % ("kernel", "excl ms", "total ms", "n/step", "avg us"))
for k, v in excl.most_common(45):
    print("%-58s %9.3f %9.3f %8.1f %9.1f" % (k, v / steps / 1e6, tot[k] / steps / 1e6, cnt[k] / steps, tot[k] / cnt[k] / 1e3))
