#!/bin/bash
# run ON the GPU box: kernel trace of a short bench (graph replay) -> tools/trace_timeline.py + trace_gaps.py -> gpurun_out/<tag>_timeline.txt
TAG=${1:-tl}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_$TAG" -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side-figures --no-roofline "$@" > "$O/${TAG}_bench.json" 2> "$O/${TAG}_trace.err"
python3 "$R/tools/trace_timeline.py" "$O/trace_$TAG" 2 > "$O/${TAG}_timeline.txt"
python3 "$R/tools/trace_gaps.py" "$O/trace_$TAG" 2 >> "$O/${TAG}_timeline.txt"
rm -rf "$O/trace_$TAG"
cat "$O/${TAG}_timeline.txt"
