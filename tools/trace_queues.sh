#!/bin/bash
# run ON the GPU box: kernel trace of a short bench, then tools/trace_queues.py (kernel time per hardware queue / stream inside one stage-4 module)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_q" -- python3 "$R/bench.py" --steps 2 --warmup 2 --no-cpu-baseline --no-side-figures --no-roofline > /dev/null 2> "$O/trace_q.err"
f=$(find "$O/trace_q" -name "*kernel_trace.csv" | head -1)
head -1 "$f"
python3 "$R/tools/trace_queues.py" "$f"; python3 "$R/tools/trace_small_grids.py" "$f"
rm -rf "$O/trace_q"
