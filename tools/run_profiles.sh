#!/bin/bash
# Produces the measurement set committed under profiles/ (run ON the GPU box, from the repo root):
#   tools/run_profiles.sh <tag>        e.g.  gpurun -- 'bash tools/run_profiles.sh r02_a'
# Per model: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, as MI355X_MICROARCH.md prescribes) of a
# 2-step bench -> pmc_traffic_<model>.json; a --kernel-trace --stats run of the bench command; the plain bench line.
# (the profiled runs use --eager: the same kernels, and the step count the per-step figures divide by stays what the command line says --
#  a graph replay adds its two warm-up steps)
# Everything lands in gpurun_out/<tag>/ (merged back by gpurun); copy what is to be judged into profiles/.
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for m in ocrnet_hrnet48 ocrnet_r50 deeplabv3plus_r50; do
  if [ "$m" != "deeplabv3plus_r50" ]; then
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$m" -- python3 "$R/bench.py" --model $m --eager --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-side-figures > "$O/pmc_fetch_$m.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$m" -- python3 "$R/bench.py" --model $m --eager --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-side-figures > "$O/pmc_write_$m.log" 2>&1
    python3 "$R/tools/pmc_traffic.py" "$O/pmc_fetch_$m" "$O/pmc_write_$m" $m > "$O/pmc_traffic_$m.json"
    mkdir -p "$R/profiles" && cp "$O/pmc_traffic_$m.json" "$R/profiles/r05_s2_pmc_traffic_$m.json"     # bench.py reads it for roofline.traffic
    rm -rf "$O"/pmc_fetch_$m "$O"/pmc_write_$m
  fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$m" -o p -- python3 "$R/bench.py" --model $m --eager --steps 3 --warmup 1 --no-cpu-baseline --no-side-figures > "$O/bench_prof_$m.json" 2> "$O/bench_prof_$m.err"
  cp $(ls "$O"/prof_$m/*/p_kernel_stats.csv "$O"/prof_$m/p_kernel_stats.csv 2>/dev/null | head -1) "$O/kernel_stats_$m.csv"
  rm -rf "$O/prof_$m"
  if [ "$m" = "ocrnet_hrnet48" ]; then
    python3 "$R/bench.py" --model $m --steps 20 --warmup 5 > "$O/bench_$m.json" 2> "$O/bench_$m.err"
    CATSEG_PRECISION=fp32 python3 "$R/bench.py" --model $m --steps 10 --warmup 3 --no-cpu-baseline --no-side-figures > "$O/bench_${m}_fp32only.json" 2> /dev/null
  else
    python3 "$R/bench.py" --model $m --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_$m.json" 2> "$O/bench_$m.err"
  fi
done
python3 "$R/bench.py" --infer > "$O/bench_infer.json" 2> "$O/bench_infer.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_infer" -o p -- python3 "$R/bench.py" --infer --steps 3 --warmup 1 > "$O/bench_prof_infer.json" 2> "$O/bench_prof_infer.err"
cp $(ls "$O"/prof_infer/*/p_kernel_stats.csv "$O"/prof_infer/p_kernel_stats.csv 2>/dev/null | head -1) "$O/kernel_stats_infer.csv"
rm -rf "$O/prof_infer"
# two ranks (gloo rendezvous) sharing the one GPU: the data-parallel path end to end (no 8-GPU node is available to the builder)
CATSEG_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 "$R/bench.py" --gpus 2 --steps 5 --warmup 2 --batch 4 --no-cpu-baseline --no-side-figures > "$O/bench_2rank_gloo_1gpu.json" 2> "$O/bench_2rank_gloo_1gpu.err"
tail -c 300 "$O"/bench_ocrnet_hrnet48.json
