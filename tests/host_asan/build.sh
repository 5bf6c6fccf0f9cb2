#!/bin/bash
# Builds libcatseg_hip_asan.so (every translation unit of csrc/ with the HOST pass under AddressSanitizer + UBSan; device code unsanitised: GPU
# ASan is not available on the target pool) and the driver next to it, then runs the driver.  CPU only; used by tests/test_host_asan_cpu.py.
# This directory is listed in .gpurunignore: nothing here is needed, or wanted, on the GPU box.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
R=$(cd "$HERE/../.." && pwd)
C=$R/miccai2021_cataract_semantic_segmentation_amd/csrc
B=$C/build_asan
CLANG=/opt/rocm/lib/llvm/bin/clang++
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
mkdir -p "$B"
SRCS=$(sed -n 's/^SRCS := //p' "$C/Makefile")
build_one() {   # incremental: rebuild an object when its source or a header is newer
  src=$1; obj=$B/$(basename "${src%.*}").o
  if [ ! -f "$obj" ] || [ -n "$(find "$C" "$R/include" -newer "$obj" \( -name '*.hip' -o -name '*.h' -o -name '*.cpp' \) -print -quit)" ]; then
    hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -I"$R/include" -Wno-comment $SAN -fno-gpu-sanitize -fno-omit-frame-pointer -c "$src" -o "$obj"
  fi
}
export -f build_one; export B C R SAN
printf '%s\n' $SRCS capi.cpp | xargs -P 8 -I{} bash -c 'build_one "$C/{}"'
OBJS=$(for f in $SRCS capi.cpp; do echo "$B/$(basename "${f%.*}").o"; done)
if [ ! -f "$B/libcatseg_hip_asan.so" ] || [ -n "$(find "$B" -name '*.o' -newer "$B/libcatseg_hip_asan.so" -print -quit)" ]; then
  hipcc --offload-arch=gfx950 -shared -fPIC $SAN -shared-libsan -o "$B/libcatseg_hip_asan.so" $OBJS
fi
RT=$(dirname "$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)")
$CLANG -std=c++17 -g -O1 -Wno-comment $SAN -shared-libsan -I"$R/include" "$HERE/driver.cpp" -o "$B/host_driver" -L"$B" -lcatseg_hip_asan \
  -Wl,-rpath,"$B" -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,"$RT"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 "$B/host_driver"
