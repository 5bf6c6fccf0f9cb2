"""prints the parts of a bench.py JSON line one reads after a change: step time, the families' event times, the HBM-bound kernels, the plan
   usage: python bench.py ... | python tools/bench_brief.py   (or: bench_brief.py <file>)"""
import json, sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
line = [l for l in src.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print("%s  %.2f frames/s  %.2f ms/step  n_gpus %d  loss %s" % (d["metric"], d["value"], d["ms_per_step"], d["n_gpus"], d["config"].get("final_loss")))
r = d.get("roofline")
if r:
    print("dominant: %s" % r["kernel"][:80])
    print("  achieved %.1f of %.1f %s = %.3f; avg launch %.3f ms; traffic %s" % (r["achieved"], r["peak"], r["unit"], r["frac"], r.get("avg_launch_ms", 0), r.get("traffic")))
    if "families_ms_per_step" in r:
        print("  families ms/step:", json.dumps(r["families_ms_per_step"]))
    for k, v in sorted(r.get("hbm_kernels", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
        print("  hbm %-16s %6.2f ms/step  %.2f of 8 TB/s  calls %4d  %6.2f GB" % (k, v["ms_per_step"], v["frac_of_8TBps"], v["calls_per_step"], v["algorithmic_GB_per_step"]))
    if "whole_step" in r:
        print("  whole step:", json.dumps(r["whole_step"]))
if d.get("comm"):
    print("comm:", json.dumps(d["comm"])[:1500])
if d["config"].get("plan"):
    print("plan non-default:", d["config"]["plan"]["non_default"])
if d.get("side_figures"):
    print("side:", json.dumps(d["side_figures"])[:600])
