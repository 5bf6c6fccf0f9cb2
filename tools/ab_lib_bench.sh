#!/bin/bash
# run ON the GPU box: the bench step with the shipped library against ab/<lib> (CATSEG_LIB), alternating, three rounds:  tools/ab_lib_bench.sh ab/libX.so
R=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2 3; do
  for lib in "" "$R/$1"; do
    ms=$(CATSEG_LIB=$lib python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-figures --no-roofline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $i ${lib:-shipped}: $ms ms/step"
  done
done
