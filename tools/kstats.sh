#!/bin/bash
# per-kernel time of a bench run:  tools/kstats.sh <tag> [bench args]   (on the GPU box) -> gpurun_out/<tag>/kernel_stats.csv + top list
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o p -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > "$O/bench.json" 2> "$O/bench.err"
F=$(ls "$O"/prof/*/p_kernel_stats.csv "$O"/prof/p_kernel_stats.csv 2>/dev/null | head -1)
cp "$F" "$O/kernel_stats.csv"
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 4.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel time per step: %.1f ms" % (tot / 1e6 / steps))
for r in rows[:32]:
    print("%-100s n=%6.0f %8.2f ms/step avg %8.1f us" % (r["Name"][:100], float(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
PY
rm -rf "$O/prof"
