// Which fp16 MFMA shape sustains more FLOP/s on THIS chip under load: v_mfma_f32_32x32x16_f16 or v_mfma_f32_16x16x32_f16?
// (MI355X_MICROARCH.md, DVFS give-back (7): bare loops on random data, same output tile per wave, the 16x16x32 loop ~1.15x.)
// One wave per SIMD (256 threads, launch_bounds(256, 1)), a 128 x 128 output tile per wave in registers (256 accumulator registers),
// operands random fp16 in registers (regs) or re-read from LDS every K-step (lds).  Prints TFLOP/s of both shapes.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip && ./mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__global__ __launch_bounds__(256, 1) void k32(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ half8 sm[2048];   // 32 KB
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) sm[i] = src[(blockIdx.x * 2048 + i) & 65535];
  __syncthreads();
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  half8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = sm[tid + 256 * i]; b[i] = sm[tid + 256 * (i + 4)]; }
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = sm[(tid + 256 * i + it * 64) & 2047]; b[i] = sm[(tid + 256 * (i + 4) + it * 64) & 2047]; }
    }
#pragma unroll
    for (int rep = 0; rep < 3; ++rep)      // three products per K-step, as the f16x2 kernels issue them
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <bool LDS>
__global__ __launch_bounds__(256, 1) void k16(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ half8 sm[4096];   // 64 KB
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) sm[i] = src[(blockIdx.x * 4096 + i) & 65535];
  __syncthreads();
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  half8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = sm[tid + 256 * i]; b[i] = sm[tid + 256 * (i + 8)]; }
  for (int it = 0; it < iters; ++it) {     // one iteration = K 32 (two K-16 steps of k32): half the iterations for equal work
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = sm[(tid + 256 * i + it * 64) & 4095]; b[i] = sm[(tid + 256 * (i + 8) + it * 64) & 4095]; }
    }
#pragma unroll
    for (int rep = 0; rep < 3; ++rep)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F>
static double run(F launch, double flop_per_launch, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 40; ++i) launch();           // ~ warm the clocks down to what the chip sustains
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 60;
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double tf = flop_per_launch * reps / (ms * 1e-3) / 1e12;
  printf("%-28s %8.3f ms/launch  %8.1f TFLOP/s (MFMA flops; /3 = fp32-equivalent %.1f)\n", name, ms / reps, tf, tf / 3);
  return tf;
}

int main() {
  const int nblk = 256 * 4;
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 8.f);
  half8* d; float* o;
  CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, nblk * 256 * 4));
  CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const int it32 = 4096, it16 = 2048;
  // per wave and iteration: k32: 48 MFMAs x 2*32*32*16 flop; k16: 192 x 2*16*16*32
  const double f32 = (double)nblk * 4 * it32 * 48 * 2.0 * 32 * 32 * 16, f16 = (double)nblk * 4 * it16 * 192 * 2.0 * 16 * 16 * 32;
  for (int round = 0; round < 2; ++round) {
    double a = run([&] { hipLaunchKernelGGL(k32<false>, dim3(nblk), dim3(256), 0, 0, d, o, it32); }, f32, "32x32x16 regs");
    double b = run([&] { hipLaunchKernelGGL(k16<false>, dim3(nblk), dim3(256), 0, 0, d, o, it16); }, f16, "16x16x32 regs");
    double c = run([&] { hipLaunchKernelGGL(k32<true>, dim3(nblk), dim3(256), 0, 0, d, o, it32); }, f32, "32x32x16 lds-fed");
    double e = run([&] { hipLaunchKernelGGL(k16<true>, dim3(nblk), dim3(256), 0, 0, d, o, it16); }, f16, "16x16x32 lds-fed");
    printf("ratio 16x16x32 / 32x32x16: regs %.3f  lds-fed %.3f\n", b / a, e / c);
  }
  return 0;
}
