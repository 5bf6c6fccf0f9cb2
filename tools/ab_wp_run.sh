#!/bin/bash
# run ON the GPU box: the backward-weight A/B builds (tools/ab_wp.sh) at several block counts
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== shipped"; python3 $R/tools/time_pl.py 4 2>&1 | grep "backward"
for f in $R/ab/libcatseg_pl_WP_*.so; do
  for b in ${WG_BLOCKS:-512 256 384}; do
    echo "== $(basename $f) blocks $b"; CATSEG_WG_BLOCKS=$b CATSEG_LIB=$f python3 $R/tools/time_pl.py 4 2>&1 | grep "backward"
  done
done
