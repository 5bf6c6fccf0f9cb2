#!/bin/bash
# run ON the GPU box: the bench step with a Python-level switch of the package flipped, alternating with the default, three rounds:
#   tools/ab_switch_bench.sh "ops.BN_BWD_FUSE=False" "engine.FUSE_BN_STATS=False" ...
R=${GRAFT_REPO_ROOT:-$PWD}
run() {
  python3 - "$1" <<PY 2>/dev/null | python3 -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
import sys
sys.path.insert(0, "$R")
setting = sys.argv[1]
sys.argv = ["bench.py", "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-side-figures", "--no-roofline"]
from miccai2021_cataract_semantic_segmentation_amd import ops, engine
if setting != "DEFAULT":
    exec(setting)
import bench
bench.main()
PY
}
for i in 1 2 3; do
  for s in "DEFAULT" "$@"; do
    echo "round $i $s: $(run "$s") ms/step"
  done
done
