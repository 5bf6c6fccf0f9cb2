R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03_a; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o p -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$O/bench_prof.json" 2> "$O/bench_prof.err"
cp $(ls "$O"/prof/*/p_kernel_stats.csv "$O"/prof/p_kernel_stats.csv 2>/dev/null | head -1) "$O/kernel_stats_ocrnet_hrnet48.csv"; rm -rf "$O/prof"
python3 - "$O/kernel_stats_ocrnet_hrnet48.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (9 steps incl. 2 instrumented):", tot / 1e6 / 9)
for r in rows[:45]:
    print("%-100s n=%6d %8.2f ms/step avg %8.1f us" % (r["Name"][:100], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6 / 9, float(r["AverageNs"]) / 1e3))
PY
