#!/bin/bash
# rocprofv3 kernel statistics of a short HRNet-W48 bench (3 + 1 steps, no side figures) -> gpurun_out/<tag>_kernel_stats.csv; prints the BatchNorm rows
# (run ON the GPU box from the repo root:  gpurun -- 'bash tools/stats_quick.sh tag')
set -u
TAG=${1:-quick}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -o p -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-side-figures --no-roofline > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err"
cp $(ls "$O"/prof_$TAG/*/p_kernel_stats.csv "$O"/prof_$TAG/p_kernel_stats.csv 2>/dev/null | head -1) "$O/${TAG}_kernel_stats.csv"
rm -rf "$O/prof_$TAG"
grep -E "bn_|Name" "$O/${TAG}_kernel_stats.csv" | cut -c1-60,100-220
tail -c 400 "$O/${TAG}_bench.json"
