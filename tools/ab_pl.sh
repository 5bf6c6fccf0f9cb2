#!/bin/bash
# differential-timing builds of the planes kernel (dconv3_pl.hip): ab/libcatseg_pl_<variant>.so, selected with CATSEG_LIB and timed by
# tools/time_pl.py.  Each variant removes one ingredient (wrong results: the time difference is its cost).  Built HERE (csrc/build does not
# travel to the GPU box).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/miccai2021_cataract_semantic_segmentation_amd/csrc
mkdir -p "$R/ab"
for v in ${AB_VARIANTS:-NO_WDMA NO_XDMA WSTAGGER NO_HWAIT NO_MFMA}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -Wno-comment -DPL_$v ${AB_EXTRA:-} -c $C/dconv3_pl.hip -o $R/ab/pl_$v.o
  OTHERS=$(ls $C/build/*.o | grep -v "/dconv3_pl.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libcatseg_pl_$v.so $OTHERS $R/ab/pl_$v.o
  rm $R/ab/pl_$v.o
done
ls $R/ab
