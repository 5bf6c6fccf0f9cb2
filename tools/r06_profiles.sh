#!/bin/bash
# Round-6 measurement set (run ON the GPU box, from the repo root):  gpurun -- 'bash tools/r06_profiles.sh r06_p'
# Per model: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs) of a 2-step bench -> pmc_traffic_<model>.json (copied to
# profiles/r06_pmc_traffic_<model>.json on the box so that the bench lines of THIS run cite it); --kernel-trace --stats of the bench command;
# the bench lines; the 2-rank gloo line (segmented graph replay); SQ counters of the trunk kernels; stage times; hipGraphLaunch host time.
set -u
TAG=${1:-r06_p}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for m in ocrnet_hrnet48 ocrnet_r50; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$m" -- python3 "$R/bench.py" --model $m --eager --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-side-figures > "$O/pmc_fetch_$m.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$m" -- python3 "$R/bench.py" --model $m --eager --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-side-figures > "$O/pmc_write_$m.log" 2>&1
  python3 "$R/tools/pmc_traffic.py" "$O/pmc_fetch_$m" "$O/pmc_write_$m" $m > "$O/pmc_traffic_$m.json"
  cp "$O/pmc_traffic_$m.json" "$R/profiles/r06_pmc_traffic_$m.json"
  rm -rf "$O"/pmc_fetch_$m "$O"/pmc_write_$m
done
for m in ocrnet_hrnet48 ocrnet_r50 deeplabv3plus_r50; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$m" -o p -- python3 "$R/bench.py" --model $m --eager --steps 3 --warmup 1 --no-cpu-baseline --no-side-figures > "$O/bench_prof_$m.json" 2> "$O/bench_prof_$m.err"
  cp $(ls "$O"/prof_$m/*/p_kernel_stats.csv "$O"/prof_$m/p_kernel_stats.csv 2>/dev/null | head -1) "$O/kernel_stats_$m.csv"
  if [ "$m" = "ocrnet_hrnet48" ]; then
    python3 "$R/tools/trace_by_grid.py" "$O/prof_$m" > "$O/trace_by_grid_$m.txt" 2>&1
    python3 "$R/tools/trace_sequence.py" "$O/prof_$m" 0 > "$O/kernel_sequence_$m.txt" 2>&1
  fi
  rm -rf "$O/prof_$m"
done
python3 "$R/bench.py" --steps 20 --warmup 5 > "$O/bench_ocrnet_hrnet48.json" 2> "$O/bench_ocrnet_hrnet48.err"
CATSEG_PRECISION=fp32 python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-side-figures > "$O/bench_ocrnet_hrnet48_fp32only.json" 2> /dev/null
python3 "$R/bench.py" --eager --steps 10 --warmup 3 --no-cpu-baseline --no-side-figures --no-roofline > "$O/bench_ocrnet_hrnet48_eager.json" 2> /dev/null
for m in ocrnet_r50 deeplabv3plus_r50; do
  python3 "$R/bench.py" --model $m --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_$m.json" 2> "$O/bench_$m.err"
done
python3 "$R/bench.py" --infer > "$O/bench_infer.json" 2> "$O/bench_infer.err"
# two ranks (gloo) sharing the one GPU through bench.py's own launcher: the data-parallel path end to end in its DEFAULT mode (segmented graph replay)
CATSEG_DIST_BACKEND=gloo python3 "$R/bench.py" --gpus 2 --steps 5 --warmup 2 --batch 4 --no-cpu-baseline --no-side-figures > "$O/bench_2rank_gloo_1gpu.json" 2> "$O/bench_2rank_gloo_1gpu.err"
CATSEG_DIST_BACKEND=gloo python3 "$R/bench.py" --gpus 2 --steps 5 --warmup 2 --batch 4 --eager --no-cpu-baseline --no-side-figures --no-roofline > "$O/bench_2rank_gloo_1gpu_eager.json" 2> /dev/null
python3 "$R/tools/stage_times.py" > "$O/stage_times.txt" 2>&1
python3 "$R/tools/host_time.py" > "$O/host_time.txt" 2>&1
bash "$R/tools/pmc_dconv3.sh" $TAG/sq_dconv3_pl_96 plfwd 8,68,120,96 > /dev/null 2>&1
bash "$R/tools/pmc_dconv3.sh" $TAG/sq_dconv3_h2_48 h2fwd 8,136,240,48 > /dev/null 2>&1
bash "$R/tools/pmc_dconv3.sh" $TAG/sq_dwgrad3_pl_96 plwgrad 8,68,120,96 > /dev/null 2>&1
tail -c 400 "$O"/bench_ocrnet_hrnet48.json
bash "$R/tools/pmc_headfuse.sh" $TAG/sq_headfuse > /dev/null 2>&1
python3 "$R/tools/time_headfuse.py" > "$O/time_headfuse.txt" 2>&1
