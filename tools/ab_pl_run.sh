#!/bin/bash
# run ON the GPU box: times every ab/libcatseg_pl_<variant>.so (tools/ab_pl.sh) and the shipped library with tools/time_pl.py
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== shipped"; python3 $R/tools/time_pl.py 4 2>&1 | grep "C=\|backward"
for f in $R/ab/libcatseg_pl_*.so; do
  echo "== $(basename $f)"; CATSEG_LIB=$f python3 $R/tools/time_pl.py 4 2>&1 | grep "C=\|backward"
done
