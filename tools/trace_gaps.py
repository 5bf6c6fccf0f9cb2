"""GPU idle time in a rocprofv3 --kernel-trace of the bench: union of kernel intervals vs wall time over the last `steps` steps
(delimited by adam_kernel), and the kernels that precede the largest share of idle gaps.  usage: trace_gaps.py <dir> [steps]"""
import collections, csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
lo, hi = adam[-steps - 1] + 1, adam[-1] + 1
ev = ev[lo:hi]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy, cur_end, gaps = 0, ev[0][0], collections.Counter()
gapn = collections.Counter()
last = None
for s, e, n in ev:
    if s > cur_end:
        if last is not None:
            key = last.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
            gaps[key] += s - cur_end
            gapn[key] += 1
        busy += 0
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        last = n
wall = t1 - t0
print("steps %d: wall %.2f ms/step, GPU busy (union of kernels) %.2f ms/step, idle %.2f ms/step (%.1f %%), %d kernels/step" % (
    steps, wall / steps / 1e6, busy / steps / 1e6, (wall - busy) / steps / 1e6, 100.0 * (wall - busy) / wall, len(ev) // steps))
print("idle time by the kernel that ended before the gap (ms/step, gaps/step, mean gap us):")
for k, v in gaps.most_common(25):
    print("  %-62s %7.3f %7.1f %7.2f" % (k, v / steps / 1e6, gapn[k] / steps, v / gapn[k] / 1e3))
