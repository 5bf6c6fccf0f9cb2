#!/bin/bash
# run ON the GPU box: bench steps of the three training models under environment settings, alternating with the default, two rounds:
#   tools/ab_env_models.sh "CATSEG_PLAN=p1_min_rows=15000" ...
R=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2; do
  for m in ocrnet_hrnet48 ocrnet_r50 deeplabv3plus_r50; do
    for setting in "DEFAULT=1" "$@"; do
      ms=$(env $setting python3 $R/bench.py --model $m --steps 12 --warmup 4 --no-cpu-baseline --no-side-figures --no-roofline 2>/dev/null | python3 -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "round $i $m $setting: $ms ms/step"
    done
  done
done
