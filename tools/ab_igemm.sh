#!/bin/bash
# address-only experiment for the fp32 implicit GEMM (run HERE): ab/libcatseg_f32_base.so and ab/libcatseg_f32_blocked.so, the second
# with -DIGEMM_BLOCKED_AB (forward operands addressed as if laid out [C/16][pixel][16] / [K/16][N][16]; values are garbage)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/miccai2021_cataract_semantic_segmentation_amd/csrc
mkdir -p "$R/ab"
cp $R/miccai2021_cataract_semantic_segmentation_amd/libcatseg_hip.so $R/ab/libcatseg_f32_base.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -Wno-comment -DIGEMM_BLOCKED_AB -c $C/igemm.hip -o $R/ab/igemm_blk.o
OTHERS=$(ls $C/build/*.o | grep -v "/igemm.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libcatseg_f32_blocked.so $OTHERS $R/ab/igemm_blk.o
rm $R/ab/igemm_blk.o
ls -la $R/ab
