// Nearest-neighbour resize (+ horizontal flip, + running mean) of NHWC tensors: the image / mask side of the test-time
// augmentation the reference runs through ttach (managers/BaseManager.py:652-660: HorizontalFlip x Scale(0.75 ... 2),
// merge 'mean').  Index rule of F.interpolate(mode='nearest', size=...):  src = min(floor(dst * (float)in / out), in - 1).
// HBM-bound: one read + one (read-modify-)write per output element.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int B,
                                                             int Hi, int Wi, int Ho, int Wo, int C, int flip, float sh, float sw,
                                                             int accumulate, float divide_by, int vec) {
  const int c4n = (C + 3) >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = (long long)B * Ho * Wo * c4n;
  if (i >= n) return;
  const int c4 = (int)(i % c4n);
  long long t = i / c4n;
  const int x = (int)(t % Wo);
  t /= Wo;
  const int y = (int)(t % Ho), b = (int)(t / Ho);
  const int xd = flip == 2 ? Wo - 1 - x : x;                 // flip of the RESULT (de-augmentation of a mask)
  int sy = min((int)floorf((float)y * sh), Hi - 1);
  int sx = min((int)floorf((float)xd * sw), Wi - 1);
  if (flip == 1) sx = Wi - 1 - sx;                            // flip of the SOURCE (augmentation of the image)
  const float* s = src + (((long long)b * Hi + sy) * Wi + sx) * lds + c4 * 4;
  float* d = dst + (((long long)b * Ho + y) * Wo + x) * ldd + c4 * 4;
  const int nc = min(4, C - c4 * 4);
  if (nc == 4 && vec) {
    float4 v = *reinterpret_cast<const float4*>(s);
    if (accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(d);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    if (divide_by != 0.f) { v.x = __fdiv_rn(v.x, divide_by); v.y = __fdiv_rn(v.y, divide_by); v.z = __fdiv_rn(v.z, divide_by); v.w = __fdiv_rn(v.w, divide_by); }
    *reinterpret_cast<float4*>(d) = v;
  } else {
    for (int c = 0; c < nc; ++c) {
      float v = s[c];
      if (accumulate) v += d[c];
      if (divide_by != 0.f) v = __fdiv_rn(v, divide_by);
      d[c] = v;
    }
  }
}

}  // namespace

extern "C" int catseg_resize_nearest(const float* src, int lds, float* dst, int ldd, int B, int Hi, int Wi, int Ho, int Wo, int C, int flip,
                                     int accumulate, float divide_by, catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "resize_nearest: bad dims");
  CS_REQUIRE(lds >= C && ldd >= C, "resize_nearest: ld must be >= C");
  // full-resolution logits are stored compact (ld = K, e.g. 25): 16-byte accesses only when both sides allow them
  const int vec = (lds % 4 == 0 && ldd % 4 == 0 && cs_aligned16(src) && cs_aligned16(dst)) ? 1 : 0;
  CS_REQUIRE(flip >= 0 && flip <= 2, "resize_nearest: flip must be 0 (none), 1 (source) or 2 (result)");
  const long long n = (long long)B * Ho * Wo * ((C + 3) / 4);
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  hipLaunchKernelGGL(resize_nearest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, B, Hi,
                     Wi, Ho, Wo, C, flip, sh, sw, accumulate, divide_by, vec);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
