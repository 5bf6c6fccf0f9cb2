#!/bin/bash
# the two --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs) of a 2-step bench for one model -> gpurun_out/<tag>/pmc_traffic_<model>.json
# usage (on the GPU box): bash tools/pmc_traffic_run.sh <tag> [model]
set -u
TAG=${1:-pmc}; m=${2:-ocrnet_hrnet48}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ARGS="--model $m --eager --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-side-figures"
if [ "$m" = "infer" ]; then ARGS="--infer --steps 1 --warmup 1 --no-roofline --no-cpu-baseline"; fi   # config 5: 2 forward steps of 4 frames
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$m" -- python3 "$R/bench.py" $ARGS > "$O/pmc_fetch_$m.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$m" -- python3 "$R/bench.py" $ARGS > "$O/pmc_write_$m.log" 2>&1
python3 "$R/tools/pmc_traffic.py" "$O/pmc_fetch_$m" "$O/pmc_write_$m" $m > "$O/pmc_traffic_$m.json"
rm -rf "$O"/pmc_fetch_$m "$O"/pmc_write_$m
python3 -c "
import json; d=json.load(open('$O/pmc_traffic_$m.json'))
for k,v in d['kernels'].items(): print(k, v['kernel_launches_per_step'], round(v['hbm_bytes_per_step']/1e9,2),'GB/step')"
