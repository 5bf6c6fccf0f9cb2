#!/bin/bash
# per-kernel time of any python tool:  tools/kstats_any.sh <tag> <script.py> [args]   (on the GPU box) -> gpurun_out/<tag>/kernel_stats.csv + top list
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
S=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o p -- python3 "$S" "$@" > "$O/out.log" 2> "$O/err.log"
F=$(ls "$O"/prof/*/p_kernel_stats.csv "$O"/prof/p_kernel_stats.csv 2>/dev/null | head -1)
cp "$F" "$O/kernel_stats.csv"
cat "$O/out.log"
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print("%-90s n=%6d %9.2f ms total avg %8.1f us" % (r["Name"][:90], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf "$O/prof"
