#!/bin/bash
# differential-timing builds of dconv3_b3_kernel (run HERE, before gpurun): ab/libcatseg_dc_<variant>.so, selected with CATSEG_LIB
# (tools/ab_run_dconv3.sh times them on the GPU box).  Each variant removes one ingredient (wrong results: the time difference is its cost).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/miccai2021_cataract_semantic_segmentation_amd/csrc
mkdir -p "$R/ab"
for v in ${AB_VARIANTS:-base NO_STASH NO_FETCH NO_DMA NO_SYNC NO_WREAD NO_XREAD NO_STORE ALL}; do
  D=""
  case $v in base) ;; ALL) D="-DDC_NO_STASH -DDC_NO_FETCH -DDC_NO_DMA -DDC_NO_SYNC -DDC_NO_WREAD -DDC_NO_XREAD -DDC_NO_STORE";; *) D="-DDC_$v";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -Wno-comment $D ${AB_EXTRA:-} -c $C/dconv3_b3.hip -o $R/ab/dc_$v.o
  OTHERS=$(ls $C/build/*.o | grep -v dconv3_b3.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libcatseg_dc_$v.so $OTHERS $R/ab/dc_$v.o
  rm $R/ab/dc_$v.o
done
ls $R/ab
