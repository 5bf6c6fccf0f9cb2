#!/bin/bash
# Produces the measurement set committed under profiles/ (run ON the GPU box, from the repo root):
#   tools/run_profiles.sh <tag>        e.g.  gpurun -- 'bash tools/run_profiles.sh r01_f'
# Per model: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md prescribes) of a
# 2-step bench -> profiles/r02_pmc_traffic_<model>.json; a --kernel-trace --stats run of the bench command; the plain bench line.
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for m in ocrnet_hrnet48 ocrnet_r50; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$m" -- python3 "$R/bench.py" --model $m --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > "$O/pmc_fetch_$m.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$m" -- python3 "$R/bench.py" --model $m --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > "$O/pmc_write_$m.log" 2>&1
  python3 "$R/tools/pmc_traffic.py" "$O/pmc_fetch_$m" "$O/pmc_write_$m" $m > "$R/profiles/r02_pmc_traffic_$m.json" && cp "$R/profiles/r02_pmc_traffic_$m.json" "$O/"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$m" -o p -- python3 "$R/bench.py" --model $m --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench_prof_$m.json" 2> "$O/bench_prof_$m.err"
  python3 "$R/bench.py" --model $m > "$O/bench_$m.json" 2> "$O/bench_$m.err"
  rm -f "$O"/pmc_*_$m/*/*kernel_trace.csv    # large, not needed once the traffic file exists
done
python3 "$R/bench.py" --infer > "$O/bench_infer.json" 2> "$O/bench_infer.err"
tail -c 400 "$O"/bench_ocrnet_hrnet48.json
