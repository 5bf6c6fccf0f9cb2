// The first convolution of the HRNet stem -- nn.Conv2d(3, 64, 3, stride 2, padding 1, bias=False) on the normalised image
// (models/HRNetv2.py:281-283 of the reference) -- forward and backward-weight, as HBM-bound direct kernels in exact fp32.
//
// K = 27: the implicit-GEMM kernels gathered it as 9 taps x 4 padded channels and ran the layer at 0.5 TB/s (642 us forward + 89 us for the
// NCHW -> NHWC-4 repack of the image, 351 us backward-weight at 8 x 3 x 544 x 960).  The layer's floor is its output: 267 MB written (forward),
// 267 MB of dy read (backward-weight); the image is 50 MB.  Here a block owns ONE output row (segments of 480 pixels for wider images): the three input rows it needs are staged
// once in LDS as [ky][column][c0 c1 c2 0] (from NCHW or NHWC-4 memory: the strides are arguments, no repack pass), and thread = (pixel lane, channel
// pair / quad): 8- / 16-byte stores / loads of y / dy (a wave covers two / four pixels x 256 bytes), one LDS read of an input column serves
// two / four channels.  Arithmetic of the forward pass: the 27 products of an output accumulated in fp64 and rounded once
// (see stem3_pixel); backward-weight: fp32 FMA chains over the pixels, fixed-order merges.
// BatchNorm statistics: (K, sum(v - K), sum((v - K)^2)) per channel with K = the block's first pixel: one partial row per block with its pixel
// count, merged by catseg_bn_finalize_counts in fp64.
#include "common.h"

namespace {

constexpr int ST_CO = 64;            // output channels
constexpr int ST_SEG = 480;          // output pixels of a row per block (wider rows: several blocks, blockIdx.y)
constexpr int ST_WP = 2 * ST_SEG + 1;                 // staged input columns per filter row
constexpr int ST_LDS = 3 * ST_WP * 16;                // bytes: [ky][column][c0 c1 c2 0]

struct Stem3Args {
  const float* x;
  long long sb, sc, sy, sx;          // element strides of the image for (batch, channel, row, column)
  int B, H, W, Ho, Wo, nseg;
  const float* w;                    // [64][3][3][3] (o, ky, kx, c): OHWI
};

// stage input rows 2 oy - 1 .. 2 oy + 1, columns 2 ox0 - 1 .. of image b: sh[ky * ST_WP + j] = x[b][0..2][2 oy - 1 + ky][2 ox0 - 1 + j] (zero outside)
__device__ __forceinline__ void stem3_stage(const Stem3Args& a, int b, int oy, int ox0, int npx, f32x4* sh) {
  const int ncol = 2 * npx + 1;
  for (int i = threadIdx.x; i < 3 * ncol; i += blockDim.x) {
    const int ky = i / ncol, j = i - ky * ncol;
    const int iy = 2 * oy - 1 + ky, ix = 2 * ox0 - 1 + j;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
      const float* p = a.x + b * a.sb + iy * a.sy + ix * a.sx;
      v[0] = p[0]; v[1] = p[a.sc]; v[2] = p[2 * a.sc];
    }
    sh[ky * ST_WP + j] = v;
  }
}

// one output pixel's two channels 2 q, 2 q + 1: the 27 products accumulated in FP64 and rounded ONCE.  The rounding error of this layer is what
// the rest of the network amplifies most (x 250 from the stem to the logits, tools/error_growth.py): a single fp32 FMA chain here -- the CPU
// path's arithmetic -- raised the full-resolution logits' RMS distance to fp64 from 6.7e-5 to 7.2e-5 and the label disagreements from 65 to 89
// (tests/test_fullres_gpu.py), the implicit GEMM's two-level fp32 chains sat in between.  fp32 x fp32 products are exact in fp64 and 27 of
// them lose nothing that survives the final rounding: the layer's output is the correctly rounded convolution.  The kernel stays HBM-bound
// (27 conversions + 54 fp64 FMAs per thread and pixel).
__device__ __forceinline__ void stem3_pixel(const f32x4* sh, int px, const double (&w)[27][2], double& a0, double& a1) {
  a0 = 0.0; a1 = 0.0;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const f32x4 v = sh[ky * ST_WP + 2 * px + kx];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double x = (double)v[c];
        a0 = __builtin_fma(x, w[(ky * 3 + kx) * 3 + c][0], a0);
        a1 = __builtin_fma(x, w[(ky * 3 + kx) * 3 + c][1], a1);
      }
    }
}

// thread = (pixel lane 0..7, channel pair 0..31): 8-byte stores (a wave covers two pixels x 256 bytes), one LDS read of an input column serves two
// channels.  BatchNorm partials: K = the block's FIRST pixel (every thread computes it for its channels), sums of (v - K) and (v - K)^2 over the
// thread's pixels, merged over the 8 pixel lanes by a lane shuffle and through LDS in a fixed order: one partial row + pixel count per block.
__global__ __launch_bounds__(256) void stem3_fwd_kernel(const Stem3Args a, const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                        float* __restrict__ part, int* __restrict__ counts) {
  __shared__ f32x4 sh[3 * ST_WP];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int b = blockIdx.x / a.Ho, oy = blockIdx.x - b * a.Ho;
  const int ox0 = blockIdx.y * ST_SEG, npx = min(ST_SEG, a.Wo - ox0);
  const int q = threadIdx.x & 31, pl = threadIdx.x >> 5, wave = threadIdx.x >> 6;
  double w[27][2];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    w[k][0] = (double)a.w[(2 * q) * 27 + k];
    w[k][1] = (double)a.w[(2 * q + 1) * 27 + k];
  }
  const double b0 = bias ? (double)bias[2 * q] : 0.0, b1 = bias ? (double)bias[2 * q + 1] : 0.0;
  stem3_stage(a, b, oy, ox0, npx, sh);
  __syncthreads();
  float* yrow = y + ((long long)blockIdx.x * a.Wo + ox0) * ldy + 2 * q;
  double a0, a1;
  stem3_pixel(sh, 0, w, a0, a1);
  const f32x2 K = {(float)(a0 + b0), (float)(a1 + b1)};
  f32x2 s1 = {0.f, 0.f}, s2 = s1;
#pragma unroll 2
  for (int px = pl; px < npx; px += 8) {
    stem3_pixel(sh, px, w, a0, a1);
    const f32x2 v = {(float)(a0 + b0), (float)(a1 + b1)};
    *(f32x2*)(yrow + (long long)px * ldy) = v;
    const f32x2 d = v - K;
    s1 += d;
    s2 += d * d;
  }
  if (part == nullptr) return;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    s1[j] += __shfl_xor(s1[j], 32, 64);
    s2[j] += __shfl_xor(s2[j], 32, 64);
  }
  __syncthreads();
  f32x4* red = sh;                   // [3 waves][32 pairs]: (s1.x, s1.y, s2.x, s2.y)
  if (wave > 0 && (threadIdx.x & 63) < 32) red[(wave - 1) * 32 + q] = f32x4{s1[0], s1[1], s2[0], s2[1]};
  __syncthreads();
  if (threadIdx.x < 32) {
    const long long prow = (long long)blockIdx.x * a.nseg + blockIdx.y;
    float* p = part + prow * 3 * ST_CO + 2 * q;
    const f32x4 t = ((f32x4{s1[0], s1[1], s2[0], s2[1]} + red[q]) + red[32 + q]) + red[64 + q];
    *(f32x2*)p = K;
    *(f32x2*)(p + ST_CO) = f32x2{t[0], t[1]};
    *(f32x2*)(p + 2 * ST_CO) = f32x2{t[2], t[3]};
    if (q == 0) counts[prow] = npx;
  }
}

// dw[o][ky][kx][c] = sum over the output pixels of dy[p][o] x[tap(p, ky, kx)][c]: a block walks the (output row, segment) items blockIdx.x,
// + gridDim.x, ...  Thread = (pixel lane 0..15, channel quad 0..15): 16-byte loads of dy (a 4-byte load per 27 FMAs left ~2 MB in flight on the
// chip: 337 us, latency-bound), 27 x 4 accumulators over its pixels, one LDS read of an input column serves four channels.  The 16 pixel lanes
// are merged by lane shuffles (the four of a wave) and through LDS (the four waves) in a fixed order; the block leaves one slab [64][27];
// stem3_wgrad_reduce_kernel sums the slabs in a fixed order.
__global__ __launch_bounds__(256) void stem3_wgrad_kernel(const Stem3Args a, const float* __restrict__ dy, int lddy, float* __restrict__ slabs) {
  __shared__ f32x4 sh[3 * ST_WP];
  static_assert(3 * ST_CO * 27 * 4 <= ST_LDS, "the wave merge reuses the staging area");
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  f32x4 acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int items = a.B * a.Ho * a.nseg;
  for (int it = blockIdx.x; it < items; it += gridDim.x) {
    const int row = it / a.nseg, seg = it - row * a.nseg;
    const int b = row / a.Ho, oy = row - b * a.Ho;
    const int ox0 = seg * ST_SEG, npx = min(ST_SEG, a.Wo - ox0);
    __syncthreads();                 // (the previous item's readers are done with the staged rows)
    stem3_stage(a, b, oy, ox0, npx, sh);
    __syncthreads();
    const float* drow = dy + ((long long)row * a.Wo + ox0) * lddy + 4 * q;
#pragma unroll 4
    for (int px = pl; px < npx; px += 16) {             // (unrolled: four 16-byte dy loads in flight per thread)
      const f32x4 g = *(const f32x4*)(drow + (long long)px * lddy);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x4 v = sh[ky * ST_WP + 2 * px + kx];
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[(ky * 3 + kx) * 3 + c][j] = __builtin_fmaf(g[j], v[c], acc[(ky * 3 + kx) * 3 + c][j]);
        }
    }
  }
  // the four pixel lanes of a wave (lane bits 4, 5)
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = acc[k][j];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      acc[k][j] = t;
    }
  __syncthreads();
  float* red = (float*)sh;           // [3 waves][64][27]: waves 1..3 hand their sums to wave 0 (fixed order)
  if (wave > 0 && (threadIdx.x & 63) < 16) {
#pragma unroll
    for (int k = 0; k < 27; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[((wave - 1) * ST_CO + 4 * q + j) * 27 + k] = acc[k][j];
  }
  __syncthreads();
  if (threadIdx.x < 16) {
#pragma unroll
    for (int k = 0; k < 27; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = 4 * q + j;
        const float t = ((acc[k][j] + red[(0 * ST_CO + o) * 27 + k]) + red[(1 * ST_CO + o) * 27 + k]) + red[(2 * ST_CO + o) * 27 + k];
        slabs[((long long)blockIdx.x * ST_CO + o) * 27 + k] = t;
      }
  }
}

// dw = sum of the slabs.  Block = 16 outputs x 16 slab lanes (lane l takes slabs l, l + 16, ... on four independent load chains: one chain per
// output was ~100 dependent L2 round trips, most of the first version's 170 us); the lanes are merged through LDS in a fixed order.
__global__ __launch_bounds__(256) void stem3_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ dw) {
  const int oo = threadIdx.x & 15, sl = threadIdx.x >> 4, out = blockIdx.x * 16 + oo;
  constexpr long long S = ST_CO * 27;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (out < S) {
    int k = sl;
    for (; k + 48 < nslabs; k += 64) {
      s0 += slabs[(long long)k * S + out];
      s1 += slabs[(long long)(k + 16) * S + out];
      s2 += slabs[(long long)(k + 32) * S + out];
      s3 += slabs[(long long)(k + 48) * S + out];
    }
    for (; k < nslabs; k += 16) s0 += slabs[(long long)k * S + out];
  }
  __shared__ float red[16][17];
  red[sl][oo] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && out < S) {
    float t = red[0][oo];
    for (int l = 1; l < 16; ++l) t += red[l][oo];
    dw[out] = t;
  }
}

void stem3_args(Stem3Args& a, const float* x, long long sb, long long sc, long long sy, long long sx, int B, int H, int W, const float* w) {
  a.x = x; a.sb = sb; a.sc = sc; a.sy = sy; a.sx = sx;
  a.B = B; a.H = H; a.W = W;
  a.Ho = (H + 2 - 3) / 2 + 1; a.Wo = (W + 2 - 3) / 2 + 1;
  a.nseg = (a.Wo + ST_SEG - 1) / ST_SEG;
  a.w = w;
}
constexpr int kWgradBlocks = 768;      // three blocks per CU (46 KB of LDS each)

}  // namespace

// 1 when catseg_stem3_fwd / _bwd_weight take the layer (64 output channels; any image of at least 2 x 2 pixels)
extern "C" int catseg_stem3_supported(int H, int W, int Cout) { return Cout == ST_CO && H >= 2 && W >= 2 ? 1 : 0; }
extern "C" int catseg_stem3_partial_rows(int B, int H, int W) {
  Stem3Args a;
  stem3_args(a, nullptr, 0, 0, 0, 0, B, H, W, nullptr);
  return B * a.Ho * a.nseg;
}
extern "C" size_t catseg_stem3_wgrad_workspace(void) { return (size_t)kWgradBlocks * ST_CO * 27 * 4; }

// y[b, oy, ox, o] = sum_{ky,kx,c} x[b, c, 2 oy - 1 + ky, 2 ox - 1 + kx] w[o, ky, kx, c] (+ bias[o]): F.conv2d(x, w, bias, stride 2, padding 1) for a
// 3-channel image and 64 output channels.  x is addressed through element strides (sb, sc, sy, sx): NCHW (C H W, H W, W, 1) or NHWC-4
// (4 H W, 1, 4 W, 4).  w: OHWI [64][3][3][3].  y: NHWC rows of ldy floats.  bn_part != NULL: catseg_stem3_partial_rows(B, H, W) partial rows
// [row][3][64] = (K, sum(v - K), sum((v - K)^2)) with their pixel counts in bn_counts, for catseg_bn_finalize_counts.
extern "C" int catseg_stem3_fwd(const float* x, long long sb, long long sc, long long sy, long long sx, int B, int H, int W, const float* w,
                                const float* bias, float* y, int ldy, float* bn_part, int* bn_counts, catseg_stream_t stream) {
  CS_REQUIRE(x && w && y && B > 0 && H >= 2 && W >= 2 && ldy >= ST_CO && ldy % 2 == 0 && (((uintptr_t)y) & 7) == 0 && (((uintptr_t)bn_part) & 7) == 0,
             "stem3 fwd: bad args (y rows, partials: 8-byte aligned)");
  CS_REQUIRE((bn_part == nullptr) == (bn_counts == nullptr), "stem3 fwd: partials and counts come together");
  Stem3Args a;
  stem3_args(a, x, sb, sc, sy, sx, B, H, W, w);
  hipLaunchKernelGGL(stem3_fwd_kernel, dim3(B * a.Ho, a.nseg), dim3(256), 0, (hipStream_t)stream, a, bias, y, ldy, bn_part, bn_counts);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// dw[o, ky, kx, c] = sum_{b, oy, ox} dy[b, oy, ox, o] x[b, c, 2 oy - 1 + ky, 2 ox - 1 + kx]  (OHWI [64][3][3][3]); workspace: catseg_stem3_wgrad_workspace()
extern "C" int catseg_stem3_bwd_weight(const float* x, long long sb, long long sc, long long sy, long long sx, int B, int H, int W, const float* dy,
                                       int lddy, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(x && dy && dw && B > 0 && H >= 2 && W >= 2 && lddy >= ST_CO && lddy % 4 == 0 && cs_aligned16(dy), "stem3 bwd_weight: bad args (dy rows: 16-byte aligned)");
  if (!workspace || workspace_bytes < catseg_stem3_wgrad_workspace()) {
    catseg_set_error("stem3 bwd_weight: workspace too small");
    return CATSEG_EWORKSPACE;
  }
  Stem3Args a;
  stem3_args(a, x, sb, sc, sy, sx, B, H, W, nullptr);
  const int items = B * a.Ho * a.nseg, blocks = items < kWgradBlocks ? items : kWgradBlocks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stem3_wgrad_kernel, dim3(blocks), dim3(256), 0, st, a, dy, lddy, (float*)workspace);
  hipLaunchKernelGGL(stem3_wgrad_reduce_kernel, dim3((ST_CO * 27 + 15) / 16), dim3(256), 0, st, (const float*)workspace, blocks, dw);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
