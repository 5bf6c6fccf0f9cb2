#!/bin/bash
# within-run A/B of ONE source file: builds ab/libcatseg_old_<stem>.so from the file as committed at <rev> (default HEAD) and the
# rest of the current objects; the working-tree library is the "new" side.  Run HERE, then time both on the GPU box with
# CATSEG_LIB=ab/libcatseg_old_<stem>.so against the default library in the same gpurun call.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/miccai2021_cataract_semantic_segmentation_amd/csrc
stem=$1; rev=${2:-HEAD}
mkdir -p "$R/ab"
make -s -C "$C" -j8
tmp=$(mktemp -d)
git -C "$R" show "$rev:miccai2021_cataract_semantic_segmentation_amd/csrc/$stem.hip" > "$tmp/$stem.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$R/include" -I"$C" -Wno-comment -c "$tmp/$stem.hip" -o "$tmp/$stem.o"
OTHERS=$(ls "$C"/build/*.o | grep -v "/$stem.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/ab/libcatseg_old_$stem.so" $OTHERS "$tmp/$stem.o"
rm -rf "$tmp"
ls -la "$R/ab/libcatseg_old_$stem.so"
