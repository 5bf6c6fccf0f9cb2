#!/bin/bash
# round-6 profile set: kernel stats of the default bench (graph replay) + a kernel trace of the eager step grouped by launch size
#   tools/r06_profile.sh <tag>
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-side-figures > "$O/bench_prof.json" 2> "$O/bench_prof.err"
cp $(ls /tmp/prof_$TAG/*/p_kernel_stats.csv /tmp/prof_$TAG/p_kernel_stats.csv 2>/dev/null | head -1) "$O/kernel_stats_ocrnet_hrnet48.csv"
python3 "$R/tools/trace_by_grid.py" /tmp/prof_$TAG > "$O/trace_by_grid.txt" 2>&1
python3 "$R/tools/trace_small_grids.py" $(ls /tmp/prof_$TAG/*/p_kernel_trace.csv /tmp/prof_$TAG/p_kernel_trace.csv 2>/dev/null | head -1) > "$O/trace_small_grids.txt" 2>&1
rm -rf /tmp/prof_$TAG
