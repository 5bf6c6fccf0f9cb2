// The first convolution of the torchvision ResNet stem -- nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False) on the normalised image
// (reference: models/OCR.py:58-61, models/DeepLabv3Plus.py:32-38 build torchvision's resnet50 / resnet101; its conv1) -- training forward
// as a direct kernel whose 147 products per output are accumulated in FP64 and rounded to fp32 ONCE.
//
// Why: the rounding error of the first layer is what the rest of the network amplifies most (csrc/stem3.hip: x 250 from the stem to the
// logits).  The implicit-GEMM route (7 x 8 taps x 4 padded channels, two-level fp32 chains) left DeepLabv3+-R50 at 1.22e-3 max |logit - CPU
// fp32| at 2 x 3 x 544 x 960 where the HRNet models, whose 3 x 3 stem got this treatment in round 5, sit at 0.88e-3.  fp32 x fp32 products are
// exact in fp64 and 147 of them lose nothing that survives the final rounding: y is the correctly rounded convolution (up to double rounding).
//
// Cost: 9.4 M fp64 FMAs per output row segment, 9.8 G per step at 8 x 3 x 544 x 960 -- 0.25 ms at the 39 T FMA/s fp64 vector rate; the layer's
// HBM floor is its 267 MB output (~50 us): this kernel is fp64-FMA-bound by design, and still no slower than the implicit GEMM it replaces
// (K = 147 padded to 224, gathered 4 channels at a time).
//
// Block (512 threads) = one output row segment of up to 240 pixels: the seven input rows it needs staged once in LDS as
// [ky][column][c0 c1 c2 0] (from NCHW or NHWC-4 memory through element strides, no repack pass), the weights once as DOUBLES [147][64].
// Thread = (channel pair q = 0..31, pixel group 0..15), four consecutive output pixels x two channels per pass: per filter row 13 LDS
// reads of input columns (broadcast across the 32 channel pairs), 21 16-byte LDS reads of weight pairs, 168 fp64 FMAs.
// BatchNorm partials as csrc/stem3.hip: (K, sum(v - K), sum((v - K)^2)) per channel with K = the block's first pixel, one partial row per
// block with its pixel count, merged by catseg_bn_finalize_counts in fp64.
#include "common.h"

namespace {

constexpr int S7_CO = 64;
constexpr int S7_SEG = 240;                      // output pixels of a row per block
constexpr int S7_WP = 2 * S7_SEG + 5;            // staged input columns per filter row
constexpr int S7_K = 147;                        // 7 x 7 x 3
constexpr int S7_LDS_X = 7 * S7_WP * 16;         // bytes
constexpr int S7_LDS_W = S7_K * S7_CO * 8;       // bytes
constexpr int S7_THREADS = 512;
constexpr int S7_PG = S7_THREADS / 32;           // pixel groups

struct Stem7Args {
  const float* x;
  long long sb, sc, sy, sx;          // element strides of the image for (batch, channel, row, column)
  int B, H, W, Ho, Wo, nseg;
  const float* w;                    // [64][7][7][3] (o, ky, kx, c): OHWI
};

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// four consecutive output pixels px0 .. px0 + 3 of the staged segment, channels 2 q and 2 q + 1
__device__ __forceinline__ void stem7_quad(const f32x4* __restrict__ shx, const f64x2* __restrict__ shw, int px0, int q, double (&acc)[4][2]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p][0] = acc[p][1] = 0.0;
#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {
    double xv[13][3];
    const f32x4* row = shx + ky * S7_WP + 2 * px0;
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const f32x4 v = row[j];
      xv[j][0] = (double)v[0]; xv[j][1] = (double)v[1]; xv[j][2] = (double)v[2];
    }
    const f64x2* wrow = shw + (ky * 21) * 32 + q;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const f64x2 wv = wrow[(kx * 3 + c) * 32];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          acc[p][0] = __builtin_fma(xv[2 * p + kx][c], wv[0], acc[p][0]);
          acc[p][1] = __builtin_fma(xv[2 * p + kx][c], wv[1], acc[p][1]);
        }
      }
  }
}

__global__ __launch_bounds__(S7_THREADS) void stem7_fwd_kernel(const Stem7Args a, const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                               float* __restrict__ part, int* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) char smem[S7_LDS_X + S7_LDS_W];      // 129.6 KB: one block (eight waves) per CU
  f32x4* shx = (f32x4*)smem;
  f64x2* shw = (f64x2*)(smem + S7_LDS_X);          // [147][32 pairs]
  const int b = blockIdx.x / a.Ho, oy = blockIdx.x - b * a.Ho;
  const int ox0 = blockIdx.y * S7_SEG, npx = min(S7_SEG, a.Wo - ox0);
  const int tid = threadIdx.x, q = tid & 31, pg = tid >> 5;
  // weights -> doubles, [k][o]
  for (int i = tid; i < S7_K * S7_CO; i += S7_THREADS) {
    const int o = i & 63, k = i >> 6;
    ((double*)shw)[k * S7_CO + o] = (double)a.w[o * S7_K + k];
  }
  // input rows 2 oy - 3 .. 2 oy + 3, columns 2 ox0 - 3 .. 2 (ox0 + npx - 1) + 3 (+ the quad overhang), zero outside the image
  {
    const int ncol = S7_WP;
    for (int i = tid; i < 7 * ncol; i += S7_THREADS) {
      const int ky = i / ncol, j = i - ky * ncol;
      const int iy = 2 * oy - 3 + ky, ix = 2 * ox0 - 3 + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
        const float* p = a.x + b * a.sb + iy * a.sy + ix * a.sx;
        v[0] = p[0]; v[1] = p[a.sc]; v[2] = p[2 * a.sc];
      }
      shx[ky * S7_WP + j] = v;
    }
  }
  __syncthreads();
  const double b0 = bias ? (double)bias[2 * q] : 0.0, b1 = bias ? (double)bias[2 * q + 1] : 0.0;
  float* yrow = y + ((long long)blockIdx.x * a.Wo + ox0) * ldy + 2 * q;
  double acc[4][2];
  f32x2 K = {0.f, 0.f};
  if (part != nullptr) {               // the shift of the statistics: the block's first pixel
    stem7_quad(shx, shw, 0, q, acc);
    K = f32x2{(float)(acc[0][0] + b0), (float)(acc[0][1] + b1)};
  }
  f32x2 s1 = {0.f, 0.f}, s2 = s1;
  for (int px0 = pg * 4; px0 < npx; px0 += S7_PG * 4) {
    stem7_quad(shx, shw, px0, q, acc);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (px0 + p < npx) {
        const f32x2 v = {(float)(acc[p][0] + b0), (float)(acc[p][1] + b1)};
        *(f32x2*)(yrow + (long long)(px0 + p) * ldy) = v;
        const f32x2 d = v - K;
        s1 += d;
        s2 += d * d;
      }
    }
  }
  if (part == nullptr) return;
  // the two pixel groups of a wave by a lane shuffle, the eight waves through LDS in a fixed order
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    s1[j] += __shfl_xor(s1[j], 32, 64);
    s2[j] += __shfl_xor(s2[j], 32, 64);
  }
  __syncthreads();
  f32x4* red = shx;                  // [7 waves][32 pairs]: (s1.x, s1.y, s2.x, s2.y)
  const int wave = tid >> 6;
  if (wave > 0 && (tid & 63) < 32) red[(wave - 1) * 32 + q] = f32x4{s1[0], s1[1], s2[0], s2[1]};
  __syncthreads();
  if (tid < 32) {
    const long long prow = (long long)blockIdx.x * a.nseg + blockIdx.y;
    float* p = part + prow * 3 * S7_CO + 2 * q;
    f32x4 t = {s1[0], s1[1], s2[0], s2[1]};
#pragma unroll
    for (int wv = 0; wv < 7; ++wv) t += red[wv * 32 + q];
    *(f32x2*)p = K;
    *(f32x2*)(p + S7_CO) = f32x2{t[0], t[1]};
    *(f32x2*)(p + 2 * S7_CO) = f32x2{t[2], t[3]};
    if (q == 0) counts[prow] = npx;
  }
}

void stem7_args(Stem7Args& a, const float* x, long long sb, long long sc, long long sy, long long sx, int B, int H, int W, const float* w) {
  a.x = x; a.sb = sb; a.sc = sc; a.sy = sy; a.sx = sx;
  a.B = B; a.H = H; a.W = W;
  a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1;
  a.nseg = (a.Wo + S7_SEG - 1) / S7_SEG;
  a.w = w;
}

}  // namespace

// 1 when catseg_stem7_fwd takes the layer (64 output channels; any image of at least 4 x 4 pixels)
extern "C" int catseg_stem7_supported(int H, int W, int Cout) { return Cout == S7_CO && H >= 4 && W >= 4 ? 1 : 0; }
extern "C" int catseg_stem7_partial_rows(int B, int H, int W) {
  Stem7Args a;
  stem7_args(a, nullptr, 0, 0, 0, 0, B, H, W, nullptr);
  return B * a.Ho * a.nseg;
}

// y[b, oy, ox, o] = sum_{ky,kx,c} x[b, c, 2 oy - 3 + ky, 2 ox - 3 + kx] w[o, ky, kx, c] (+ bias[o]): F.conv2d(x, w, bias, stride 2, padding 3) for a
// 3-channel image and 64 output channels, every output accumulated in fp64 and rounded once.  x is addressed through element strides
// (sb, sc, sy, sx): NCHW (C H W, H W, W, 1) or NHWC-4 (4 H W, 1, 4 W, 4).  w: OHWI [64][7][7][3].  y: NHWC rows of ldy floats.
// bn_part != NULL: catseg_stem7_partial_rows(B, H, W) partial rows [row][3][64] = (K, sum(v - K), sum((v - K)^2)) with their pixel counts in
// bn_counts, for catseg_bn_finalize_counts.
extern "C" int catseg_stem7_fwd(const float* x, long long sb, long long sc, long long sy, long long sx, int B, int H, int W, const float* w,
                                const float* bias, float* y, int ldy, float* bn_part, int* bn_counts, catseg_stream_t stream) {
  CS_REQUIRE(x && w && y && B > 0 && H >= 4 && W >= 4 && ldy >= S7_CO && ldy % 2 == 0 && (((uintptr_t)y) & 7) == 0 && (((uintptr_t)bn_part) & 7) == 0,
             "stem7 fwd: bad args (y rows, partials: 8-byte aligned)");
  CS_REQUIRE((bn_part == nullptr) == (bn_counts == nullptr), "stem7 fwd: partials and counts come together");
  Stem7Args a;
  stem7_args(a, x, sb, sc, sy, sx, B, H, W, w);
  hipLaunchKernelGGL(stem7_fwd_kernel, dim3(B * a.Ho, a.nseg), dim3(S7_THREADS), 0, (hipStream_t)stream, a, bias, y, ldy, bn_part, bn_counts);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
