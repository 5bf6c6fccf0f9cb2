#!/bin/bash
# run ON the GPU box: kernel trace of a short bench -> tools/trace_sequence.py -> gpurun_out/<tag>_sequence.txt
TAG=${1:-seq}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "/tmp/trace_$TAG" -- python3 "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-side-figures --no-roofline "$@" > "$O/${TAG}_bench.json" 2> "$O/${TAG}_trace.err"
python3 "$R/tools/trace_sequence.py" "/tmp/trace_$TAG" > "$O/${TAG}_sequence.txt"
tail -3 "$O/${TAG}_sequence.txt"
