#!/bin/bash
# step-level A/B of the execution modes on one box: hipGraph replay against the eager launch loop, alternating rounds
# (round 5 also ran the asynchronous head backward-weight here: measured, no gain, removed in round 6 -- docs/DESIGN_history.md)
# usage (on the GPU box): bash tools/ab_modes.sh <tag> [rounds]
TAG=${1:-modes}; N=${2:-2}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p "$O"
A="--steps 15 --warmup 4 --no-cpu-baseline --no-side-figures --no-roofline"
for r in $(seq 1 $N); do
  python3 "$R/bench.py" $A > "$O/graph_$r.json" 2> "$O/graph_$r.err"
  python3 "$R/bench.py" $A --eager > "$O/eager_$r.json" 2> /dev/null
done
python3 - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); print("%-24s %.2f ms  %.2f frames/s  loss %.7f" % (os.path.basename(f), d["ms_per_step"], d["value"], d["config"]["final_loss"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
