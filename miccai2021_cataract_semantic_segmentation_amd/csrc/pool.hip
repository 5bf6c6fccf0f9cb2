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

// F.max_pool2d(x, 2) (kernel 2, stride 2, no padding, floor: models/FCN.py:44-53 of the reference) on NHWC fp32; idx = position of the
// maximum inside its window (0 .. 3, the FIRST maximum in row-major order, as ATen's `val > max || isnan(val)` scan keeps it)
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, unsigned char* __restrict__ idx, int B,
                                    int H, int W, int C, int Ho, int Wo) {
  const int cpt = C >> 2;
  const long long total = (long long)B * Ho * Wo * cpt;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long p = t / cpt;
    const int c = (int)(t - p * cpt) * 4;
    const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
    f32x4 best = *(const f32x4*)(x + (((long long)b * H + 2 * oy) * W + 2 * ox) * ldx + c);
    unsigned char bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const f32x4 v = *(const f32x4*)(x + (((long long)b * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * ldx + c);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; bi[j] = (unsigned char)k; }
    }
    *(f32x4*)(y + p * ldy + c) = best;
    *(uchar4*)(idx + p * C + c) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
  }
}
__global__ void maxpool2_bwd_kernel(const float* __restrict__ dy, int lddy, const unsigned char* __restrict__ idx, float* __restrict__ dx, int lddx,
                                    int B, int H, int W, int C, int Ho, int Wo) {
  const int cpt = C >> 2;
  const long long total = (long long)B * H * W * cpt;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long p = t / cpt;
    const int c = (int)(t - p * cpt) * 4;
    const int xx = (int)(p % W), yy = (int)((p / W) % H), b = (int)(p / ((long long)W * H));
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const int oy = yy >> 1, ox = xx >> 1;
    if (oy < Ho && ox < Wo) {
      const long long q = ((long long)b * Ho + oy) * Wo + ox;
      const f32x4 d = *(const f32x4*)(dy + q * lddy + c);
      const uchar4 i4 = *(const uchar4*)(idx + q * C + c);
      const int k = ((yy & 1) << 1) | (xx & 1);
      g[0] = i4.x == k ? d[0] : 0.f; g[1] = i4.y == k ? d[1] : 0.f; g[2] = i4.z == k ? d[2] : 0.f; g[3] = i4.w == k ? d[3] : 0.f;
    }
    *(f32x4*)(dx + p * lddx + c) = g;
  }
}
// out[r][0 .. C) = bias[0 .. C) (the rows ConvTranspose2d's bias initialises before the transposed convolution accumulates into them)
__global__ void bias_rows_kernel(const float* __restrict__ bias, float* __restrict__ out, int ld, long long rows, int C) {
  const long long total = rows * C;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / C;
    const int c = (int)(t - r * C);
    out[r * ld + c] = bias[c];
  }
}

}  // namespace

extern "C" int catseg_maxpool2x2_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx, int B, int H, int W, int C, catseg_stream_t stream) {
  const int Ho = H / 2, Wo = W / 2;
  CS_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && cs_aligned16(x) && cs_aligned16(y) && idx,
             "maxpool2x2 fwd: bad args");
  long long blocks = ((long long)B * Ho * Wo * (C / 4) + 255) / 256;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, idx,
                     B, H, W, C, Ho, Wo);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_maxpool2x2_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx, int B, int H, int W, int C,
                                     catseg_stream_t stream) {
  const int Ho = H / 2, Wo = W / 2;
  CS_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C % 4 == 0 && lddx % 4 == 0 && lddy % 4 == 0 && cs_aligned16(dx) && cs_aligned16(dy) && idx,
             "maxpool2x2 bwd: bad args");
  long long blocks = ((long long)B * H * W * (C / 4) + 255) / 256;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, dy, lddy, idx, dx,
                     lddx, B, H, W, C, Ho, Wo);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_bias_rows(const float* bias, float* out, int ld, long long rows, int C, catseg_stream_t stream) {
  CS_REQUIRE(bias && out && rows > 0 && C > 0 && ld >= C, "bias_rows: bad args");
  long long blocks = (rows * C + 255) / 256;
  hipLaunchKernelGGL(bias_rows_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, bias, out, ld, rows, C);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

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
