#!/bin/bash
# differential-timing builds of the f16x2 direct kernel (dconv3_f16x2.hip): ab/libcatseg_h2_<variant>.so, selected with CATSEG_LIB and
# timed by tools/time_d3h.py (cache-cold inputs).  Each variant removes one ingredient (wrong results: the time difference is its cost).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/miccai2021_cataract_semantic_segmentation_amd/csrc
mkdir -p "$R/ab"
for v in ${AB_VARIANTS:-base NO_STASH NO_FETCH NO_DMA NO_SYNC NO_WREAD NO_XREAD NO_STORE}; do
  D=""
  case $v in base) ;; *) D="-DDC_$v";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -Wno-comment $D ${AB_EXTRA:-} -c $C/dconv3_f16x2.hip -o $R/ab/h2_$v.o
  OTHERS=$(ls $C/build/*.o | grep -v "/dconv3_f16x2.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libcatseg_h2_$v.so $OTHERS $R/ab/h2_$v.o
  rm $R/ab/h2_$v.o
done
ls $R/ab
