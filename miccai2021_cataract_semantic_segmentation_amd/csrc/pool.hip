// nn.AdaptiveAvgPool2d(S) forward / backward on NHWC fp32 (UPerNet pyramid pooling,
// models/UPerNet.py:25-33,114-118 of the reference).  Bin i covers [floor(i*H/S), ceil((i+1)*H/S)).
#include "common.h"

namespace {

__device__ __forceinline__ int bin_lo(int i, int n, int s) { return (i * n) / s; }
__device__ __forceinline__ int bin_hi(int i, int n, int s) { return ((i + 1) * n + s - 1) / s; }

// grid (S*S, B, ceil(C/256)); block 256 = channels
__global__ void aap_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int H, int W, int C, int S) {
  const int c = blockIdx.z * 256 + threadIdx.x;
  if (c >= C) return;
  const int b = blockIdx.y, i = blockIdx.x / S, j = blockIdx.x - i * S;
  const int y0 = bin_lo(i, H, S), y1 = bin_hi(i, H, S), x0 = bin_lo(j, W, S), x1 = bin_hi(j, W, S);
  float s = 0.f;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) s += x[(((long long)b * H + yy) * W + xx) * ldx + c];
  y[(((long long)b * S + i) * S + j) * C + c] = s / (float)((y1 - y0) * (x1 - x0));
}

// dx[b, yy, xx, c] (+)= sum over bins containing (yy, xx) of dy / binsize
__global__ void aap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int lddx, int B, int H, int W, int C, int S, int acc) {
  const int cpt = C >> 2;
  const long long total = (long long)B * H * W * cpt;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long p = t / cpt;
    const int c = (int)(t - p * cpt) * 4;
    const int xx = (int)(p % W), yy = (int)((p / W) % H), b = (int)(p / ((long long)W * H));
    f32x4 g = {0, 0, 0, 0};
    // every bin containing (yy, xx) contributes (bins overlap when H % S != 0 and repeat when S > H)
    for (int i = 0; i < S; ++i) {
      const int y0 = bin_lo(i, H, S), y1 = bin_hi(i, H, S);
      if (yy < y0 || yy >= y1) continue;
      for (int j = 0; j < S; ++j) {
        const int x0 = bin_lo(j, W, S), x1 = bin_hi(j, W, S);
        if (xx < x0 || xx >= x1) continue;
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
        g += *(const f32x4*)(dy + (((long long)b * S + i) * S + j) * C + c) * inv;
      }
    }
    f32x4* d = (f32x4*)(dx + p * lddx + c);
    *d = acc ? (*d + g) : g;
  }
}

}  // namespace

extern "C" int catseg_adaptive_avgpool_fwd(const float* x, int ldx, float* y, int B, int H, int W, int C, int S,
                                           catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && S > 0 && S <= 64 && ldx >= C, "adaptive avgpool: bad args");
  hipLaunchKernelGGL(aap_fwd_kernel, dim3(S * S, B, (C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, y, H, W, C, S);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_adaptive_avgpool_bwd(const float* dy, float* dx, int lddx, int B, int H, int W, int C, int S,
                                           int accumulate, catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && C % 4 == 0 && lddx % 4 == 0 && S > 0 && S <= 64 && cs_aligned16(dy) && cs_aligned16(dx), "adaptive avgpool bwd: bad args");
  long long blocks = ((long long)B * H * W * (C / 4) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(aap_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, dx, lddx, B, H, W, C, S, accumulate);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
