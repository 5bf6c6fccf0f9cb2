// The PIL-based training augmentations of the reference's image pipeline on uint8 NHWC batches [B][H][W][3], bit-exact with
// Pillow (tests/golden/augment.npz is generated with Pillow itself; oracle/augment.py is the numpy restatement):
//   * BlurPIL (utils/transforms.py:242-251): ImageFilter.GaussianBlur(radius) = 3 extended box-blur passes per direction
//     (libImaging/BoxBlur.c: 8.24 fixed point window weight ww, fractional edge weight fw, edge pixels replicated);
//   * torchvision ColorJitter on PIL images (wired at utils/utils.py:415-417): brightness / contrast / saturation =
//     Image.blend(degenerate, img, factor) with degenerate = black / rounded mean luma / luma image; hue = HSV round trip with
//     a uint8 wrap-around shift of H (libImaging/Blend.c, Convert.c);
//   * the uint8 flip + reflect pad that precedes them in the reference's order (FlipNP, PadNP; utils/utils.py:394-401).
// HBM-bound byte work: every kernel is one pass over a 12 MB batch (8 x 544 x 960 x 3).
#include "common.h"

// Pillow is plain x86-64 code: a * b + c is two roundings there.  Contraction is switched off for this file, and the one place
// where it matters (Image.blend) additionally pins the product behind an empty asm: neither __fmul_rn / __fadd_rn nor the pragma
// alone kept the compiler from emitting v_fmac_f32 there (contrast 2/3 came out one grey level low on 341 of 9657 bytes).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one ImagingLineBoxBlur8 pass along `axis` (0: rows / H, 1: columns / W).  Closed form of BoxBlur.c's running sums:
//   out = (acc * ww + (far_left + far_right) * fw + 2^23) >> 24 in uint32 arithmetic, acc = the 2 radius + 1 window pixels,
//   far = the two pixels just outside the window, indices clamped to the line
__global__ __launch_bounds__(256) void aug_box_blur_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H, int W,
                                                           int axis, const int* __restrict__ radius, const unsigned* __restrict__ ww,
                                                           const unsigned* __restrict__ fw) {
  const long long per = (long long)H * W * 3;
  const long long total = per * B;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const long long rem = i - (long long)b * per;
    const int r = radius[b];
    if (r < 0) {   // this image is not blurred
      out[i] = in[i];
      continue;
    }
    const int c = (int)(rem % 3);
    const int x = (int)((rem / 3) % W), y = (int)(rem / (3LL * W));
    const int n = axis ? W : H, pos = axis ? x : y;
    const long long stride = axis ? 3 : 3LL * W;
    const uint8_t* line = in + (long long)b * per + (axis ? (long long)y * W * 3 : (long long)x * 3) + c;
    unsigned acc = 0;
    for (int j = -r; j <= r; ++j) acc += line[clampi(pos + j, 0, n - 1) * stride];
    const unsigned far = (unsigned)line[clampi(pos - r - 1, 0, n - 1) * stride] + (unsigned)line[clampi(pos + r + 1, 0, n - 1) * stride];
    const unsigned bulk = acc * ww[b] + far * fw[b];
    out[i] = (uint8_t)((bulk + (1u << 23)) >> 24);
  }
}

__device__ __forceinline__ unsigned luma_u8(unsigned r, unsigned g, unsigned b) { return (r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16; }

// per-image sum of the luma (ImageStat.Stat(img.convert("L")).sum): integer, order-independent
__global__ __launch_bounds__(256) void aug_luma_sum_kernel(const uint8_t* __restrict__ img, int B, long long pixels, unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y;
  const uint8_t* p = img + (long long)b * pixels * 3;
  unsigned long long s = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pixels; i += (long long)gridDim.x * blockDim.x)
    s += luma_u8(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
  __shared__ unsigned long long sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&sums[b], sh[0]);
}

// Image.blend(in1, in2, alpha) per byte (Blend.c): float arithmetic WITHOUT contraction (the host library is plain x86-64 code),
// truncated to uint8; clipped when extrapolating
__device__ __forceinline__ uint8_t blend_u8(int in1, int in2, float alpha) {
  float prod = alpha * (float)(in2 - in1);
  asm volatile("" : "+v"(prod));            // the product is rounded to float before the add (no v_fmac)
  const float t = (float)in1 + prod;
  if (alpha >= 0.f && alpha <= 1.0f) return (uint8_t)(int)t;
  if (t <= 0.f) return 0;
  if (t >= 255.f) return 255;
  return (uint8_t)(int)t;
}

__device__ __forceinline__ int clip8(long long v) { return v < 0 ? 0 : (v > 255 ? 255 : (int)v); }

// Convert.c rgb2hsv_row / hsv2rgb (after colorsys.py): float variables, double literals
__device__ __forceinline__ void rgb2hsv_u8(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) {
    uh = 0;
    us = 0;
    return;
  }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float h;
  if (r == maxc) h = __fsub_rn(bc, gc);
  else if (g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
  else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
  h = (float)fmod(__dadd_rn(__ddiv_rn((double)h, 6.0), 1.0), 1.0);
  uh = clip8((long long)__dmul_rn((double)h, 255.0));
  us = clip8((long long)__dmul_rn((double)s, 255.0));
}

__device__ __forceinline__ void hsv2rgb_u8(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) {
    r = g = b = v;
    return;
  }
  const double hd = __ddiv_rn(__dmul_rn((double)(float)h, 6.0), 255.0);
  const int i = (int)floor(hd);
  const double f = (double)(float)__dsub_rn(hd, (double)(float)i);
  const double fs = (double)(float)__ddiv_rn((double)(float)s, 255.0);
  const double vd = (double)(float)v;
  const int p = clip8((long long)round(__dmul_rn(vd, __dsub_rn(1.0, fs))));
  const int q = clip8((long long)round(__dmul_rn(vd, __dsub_rn(1.0, __dmul_rn(fs, f)))));
  const int t = clip8((long long)round(__dmul_rn(vd, __dsub_rn(1.0, __dmul_rn(fs, __dsub_rn(1.0, f))))));
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// one operation of torchvision.transforms.functional per image: op[b] = 0 brightness, 1 contrast, 2 saturation, 3 hue (factor = the
// integer H shift), < 0 none
__global__ __launch_bounds__(256) void aug_color_kernel(uint8_t* __restrict__ img, int B, long long pixels, const int* __restrict__ op,
                                                        const float* __restrict__ factor, const unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y;
  const int o = op[b];
  if (o < 0) return;
  const float f = factor[b];
  uint8_t* p = img + (long long)b * pixels * 3;
  int mean = 0;
  if (o == 1) mean = (int)((double)sums[b] / (double)pixels + 0.5);   // int(ImageStat.Stat(L).mean[0] + 0.5)
  int shift = 0;
  if (o == 3) shift = ((int)f) & 0xFF;   // hue: factor[b] holds the H shift itself, int(hue_factor * 255) computed in double on the host (np.uint8(hue_factor * 255))
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pixels; i += (long long)gridDim.x * blockDim.x) {
    int r = p[i * 3], g = p[i * 3 + 1], bb = p[i * 3 + 2];
    if (o == 0) {
      r = blend_u8(0, r, f); g = blend_u8(0, g, f); bb = blend_u8(0, bb, f);
    } else if (o == 1) {
      r = blend_u8(mean, r, f); g = blend_u8(mean, g, f); bb = blend_u8(mean, bb, f);
    } else if (o == 2) {
      const int L = (int)luma_u8(r, g, bb);
      r = blend_u8(L, r, f); g = blend_u8(L, g, f); bb = blend_u8(L, bb, f);
    } else {
      int h, s, v;
      rgb2hsv_u8(r, g, bb, h, s, v);
      h = (h + shift) & 0xFF;
      hsv2rgb_u8(h, s, v, r, g, bb);
    }
    p[i * 3] = (uint8_t)r; p[i * 3 + 1] = (uint8_t)g; p[i * 3 + 2] = (uint8_t)bb;
  }
}

// uint8 -> uint8: FlipNP (bit 0 horizontal, bit 1 vertical) then PadNP(ver = (top, bottom), mode 'reflect')
__global__ __launch_bounds__(256) void aug_pad_flip_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int H, int W, int C,
                                                           const int* __restrict__ flips, int top, int bottom) {
  const int Ho = H + top + bottom;
  const long long total = (long long)B * Ho * W * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int x = (int)((i / C) % W);
    const int y = (int)((i / ((long long)C * W)) % Ho);
    const int b = (int)(i / ((long long)C * W * Ho));
    int sy = y - top;                       // reflect (no edge repeat): -1 -> 1, H -> H - 2
    if (sy < 0) sy = -sy;
    if (sy >= H) sy = 2 * (H - 1) - sy;
    const int fl = flips ? flips[b] : 0;
    if (fl & 2) sy = H - 1 - sy;
    const int sx = (fl & 1) ? W - 1 - x : x;
    out[i] = in[(((long long)b * H + sy) * W + sx) * C + c];
  }
}

int grid_of(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int catseg_aug_box_blur(const uint8_t* in, uint8_t* out, int B, int H, int W, int axis, const int32_t* radius, const uint32_t* ww,
                                   const uint32_t* fw, catseg_stream_t stream) {
  CS_REQUIRE(in && out && in != out && B > 0 && H > 0 && W > 0 && (axis == 0 || axis == 1) && radius && ww && fw, "aug_box_blur: bad args");
  hipLaunchKernelGGL(aug_box_blur_kernel, dim3(grid_of((long long)B * H * W * 3)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, axis,
                     (const int*)radius, (const unsigned*)ww, (const unsigned*)fw);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_aug_color_op(uint8_t* img, int B, int H, int W, const int32_t* op, const float* factor, void* workspace,
                                   size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(img && B > 0 && H > 0 && W > 0 && op && factor, "aug_color_op: bad args");
  if (workspace_bytes < (size_t)B * 8 || !workspace) {
    catseg_set_error("aug_color_op: workspace %zu < %zu", workspace_bytes, (size_t)B * 8);
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long pixels = (long long)H * W;
  unsigned long long* sums = (unsigned long long*)workspace;
  if (hipMemsetAsync(sums, 0, (size_t)B * 8, st) != hipSuccess) { catseg_set_error("aug_color_op: memset failed"); return CATSEG_EHIP; }
  const int gx = (int)((pixels + 255) / 256 < 512 ? (pixels + 255) / 256 : 512);
  hipLaunchKernelGGL(aug_luma_sum_kernel, dim3(gx, B), dim3(256), 0, st, (const uint8_t*)img, B, pixels, sums);
  hipLaunchKernelGGL(aug_color_kernel, dim3(gx, B), dim3(256), 0, st, img, B, pixels, (const int*)op, factor, (const unsigned long long*)sums);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_aug_pad_flip_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, const int32_t* flips, int pad_top,
                                      int pad_bottom, catseg_stream_t stream) {
  CS_REQUIRE(in && out && in != out && B > 0 && H > 1 && W > 0 && C > 0 && pad_top >= 0 && pad_bottom >= 0 && pad_top < H && pad_bottom < H,
             "aug_pad_flip: bad args (reflect padding needs pad < H)");
  hipLaunchKernelGGL(aug_pad_flip_kernel, dim3(grid_of((long long)B * (H + pad_top + pad_bottom) * W * C)), dim3(256), 0, (hipStream_t)stream, in, out,
                     B, H, W, C, (const int*)flips, pad_top, pad_bottom);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
