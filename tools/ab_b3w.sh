#!/bin/bash
# differential-timing builds of igemm_b3w_kernel (run HERE, before gpurun): ab/libcatseg_<variant>.so, selected with CATSEG_LIB
# (tools/ab_run.sh times them on the GPU box).  Variants: base, NO_PREP / NO_DMA / NO_READS / NO_SYNC / ALL (one ingredient of the K loop
# removed: wrong results, the time difference is its cost), GEN_BLOCKED (8-wave tiles with blocked-plane addresses, address-only).
# The blocked layout itself was found this way (-DB3X_BLOCKED_A / _B address-only builds, since folded into igemm_b3w_kernel<true>).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/miccai2021_cataract_semantic_segmentation_amd/csrc
mkdir -p "$R/ab"
for v in ${AB_VARIANTS:-base NO_PREP NO_DMA NO_READS NO_SYNC ALL}; do
  D=""
  case $v in base) ;; ALL) D="-DB3X_NO_PREP -DB3X_NO_DMA -DB3X_NO_READS -DB3X_NO_SYNC";; GEN_BLOCKED) D="-DB3G_BLOCKED";; *) D="-DB3X_$v";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -Wno-comment $D -c $C/igemm_bf16x3.hip -o $R/ab/b3_$v.o
  OTHERS=$(ls $C/build/*.o | grep -v igemm_bf16x3.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libcatseg_$v.so $OTHERS $R/ab/b3_$v.o
  rm $R/ab/b3_$v.o
done
ls -la $R/ab
