#!/bin/bash
# run ON the GPU box: kernel trace of a short bench -> tools/trace_gaps.py -> gpurun_out/<tag>_gpu_idle_gaps.txt
TAG=${1:-gaps}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_$TAG" -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side-figures --no-roofline > /dev/null 2> "$O/${TAG}_trace.err"
python3 "$R/tools/trace_gaps.py" "$O/trace_$TAG" 2 > "$O/${TAG}_gpu_idle_gaps.txt"
rm -rf "$O/trace_$TAG"
cat "$O/${TAG}_gpu_idle_gaps.txt"
