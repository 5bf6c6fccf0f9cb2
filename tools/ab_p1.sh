#!/bin/bash
# step-level A/B of the pointwise split-precision route (CATSEG_P1=0/1, CATSEG_P1_OPS), alternating rounds, graph replay
TAG=${1:-p1}; N=${2:-2}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p "$O"
A="--steps 15 --warmup 4 --no-cpu-baseline --no-side-figures --no-roofline"
for r in $(seq 1 $N); do
  python3 "$R/bench.py" $A > "$O/p1_on_$r.json" 2> "$O/p1_on_$r.err"
  CATSEG_P1=0 python3 "$R/bench.py" $A > "$O/p1_off_$r.json" 2> /dev/null
  CATSEG_P1_OPS=fwd,dgrad python3 "$R/bench.py" $A > "$O/p1_nowgrad_$r.json" 2> /dev/null
done
python3 - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/p1_*.json")):
    try:
        d = json.load(open(f)); print("%-24s %.2f ms  %.2f frames/s  loss %.7f" % (os.path.basename(f), d["ms_per_step"], d["value"], d["config"]["final_loss"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
