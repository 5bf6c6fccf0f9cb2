#!/bin/bash
# stem7 (fp64-accumulated ResNet stem, csrc/stem7.hip) on / off: step time of the ResNet-50 models, alternating;  usage: tools/ab_stem7.sh <tag>
TAG=${1:-stem7}; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p "$O"
A="--steps 12 --warmup 3 --no-cpu-baseline --no-side-figures --no-roofline"
for r in 1 2; do
  for m in deeplabv3plus_r50 ocrnet_r50; do
    CATSEG_STEM7=1 python3 "$R/bench.py" --model $m $A > "$O/${m}_on_$r.json" 2> /dev/null
    CATSEG_STEM7=0 python3 "$R/bench.py" --model $m $A > "$O/${m}_off_$r.json" 2> /dev/null
  done
done
python3 - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print("%-32s %.2f ms  %.2f frames/s  loss %.7f  plan %s" % (os.path.basename(f), d["ms_per_step"], d["value"], d["config"]["final_loss"], d["config"]["plan"]["non_default"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
