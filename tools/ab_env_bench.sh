#!/bin/bash
# run ON the GPU box: the bench step under environment settings, alternating with the default, three rounds:
#   tools/ab_env_bench.sh "CATSEG_WG_BLOCKS=384" "CATSEG_WG_BLOCKS=640" ...
R=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2 3; do
  for setting in "DEFAULT=1" "$@"; do
    ms=$(env $setting python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-figures --no-roofline 2>/dev/null | python3 -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $i $setting: $ms ms/step"
  done
done
