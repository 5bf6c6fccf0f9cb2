#!/bin/bash
# differential-timing builds of the planes backward-weight kernel (dwgrad3_pl.hip): ab/libcatseg_wp_<variant>.so, timed by tools/time_pl.py
# through CATSEG_LIB (tools/ab_pl_run.sh).  Built HERE (csrc/build does not travel to the GPU box).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/miccai2021_cataract_semantic_segmentation_amd/csrc
mkdir -p "$R/ab"
for v in ${AB_VARIANTS:-NO_MFMA NO_DMA NO_STORE NO_DSREAD}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -Wno-comment -DWP_$v ${AB_EXTRA:-} -c $C/dwgrad3_pl.hip -o $R/ab/wp_$v.o
  OTHERS=$(ls $C/build/*.o | grep -v "/dwgrad3_pl.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libcatseg_pl_WP_$v.so $OTHERS $R/ab/wp_$v.o
  rm $R/ab/wp_$v.o
done
ls $R/ab
