#!/bin/bash
# kernel trace of a short bench run + idle-gap analysis:  tools/ktrace.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace_$TAG -o p -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline "$@" > "$O/bench.json" 2> "$O/bench.err"
python3 "$R/tools/trace_gaps.py" /tmp/ktrace_$TAG 2 | tee "$O/gaps.txt"
python3 "$R/tools/trace_neighbors.py" /tmp/ktrace_$TAG copyBuffer | tee "$O/copybuffer_neighbors.txt"
