// Raw frame -> network input on device (reference: datasets/Dataset_from_df.py:31-69 runs this per frame on the
// host main thread, num_workers = 0 for the repeat-factor loader, managers/BaseManager.py:391):
//   label:  remap_mask(lbl, CLASS_INFO[exp][0], to_network=True)    utils/utils.py:23-47      -> 256-entry LUT
//   both:   FlipNP (vertical / horizontal, the same for image and label)   utils/transforms.py:222-240
//           PadNP(ver=(2,2), hor=(0,0), 'reflect')                          utils/transforms.py:8-20, utils/utils.py:394-401
//   image:  ToTensor (u8 HWC -> f32 CHW, / 255), optional Normalize(mean, std)  utils/utils.py:440-447
// One thread per output pixel; HBM-bound: 4 B read, 12 (+16) + 8 B written per pixel.
#include "common.h"

namespace {

__device__ __forceinline__ int reflect(int r, int n) {  // np.pad(mode='reflect'): no edge repeat
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  r = r % period;
  if (r < 0) r += period;
  return r < n ? r : period - r;
}

__global__ __launch_bounds__(256) void ingest_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lbl, int B, int H, int W,
                                                     const uint8_t* __restrict__ lut, const int32_t* __restrict__ flips, int pad_top,
                                                     int Ho, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                     float* __restrict__ x_nchw, float* __restrict__ x_nhwc4, int64_t* __restrict__ labels) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = (long long)B * Ho * W;
  if (i >= n) return;
  const int c = (int)(i % W);
  const long long t = i / W;
  const int r = (int)(t % Ho), b = (int)(t / Ho);
  const int f = flips ? flips[b] : 0;
  int sy = reflect(r - pad_top, H);      // row of the flipped frame
  int sx = c;
  if (f & 2) sy = H - 1 - sy;
  if (f & 1) sx = W - 1 - sx;
  const long long src = ((long long)b * H + sy) * W + sx;
  if (img) {
    float v[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float q = __fdiv_rn((float)img[src * 3 + ch], 255.0f);  // ToTensor: correctly rounded division, as torch's .div(255)
      if (mean) {
        q = __fsub_rn(q, mean[ch]);                       // Normalize: sub_ then div_, two roundings, no contraction
        q = __fdiv_rn(q, stdv[ch]);
      }
      v[ch] = q;
    }
    if (x_nchw) {
      const long long plane = (long long)Ho * W;
      float* o = x_nchw + (long long)b * 3 * plane + (long long)r * W + c;
      o[0] = v[0]; o[plane] = v[1]; o[2 * plane] = v[2];
    }
    if (x_nhwc4) *reinterpret_cast<float4*>(x_nhwc4 + i * 4) = make_float4(v[0], v[1], v[2], 0.f);
  }
  if (lbl) labels[i] = (int64_t)(lut ? lut[lbl[src]] : lbl[src]);
}

}  // namespace

extern "C" int catseg_ingest_u8(const uint8_t* img, const uint8_t* lbl, int B, int H, int W, const uint8_t* lut, const int32_t* flips,
                                int pad_top, int pad_bottom, const float* mean, const float* stdv, float* x_nchw, float* x_nhwc4,
                                int64_t* labels, catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && pad_top >= 0 && pad_bottom >= 0, "ingest: bad dims");
  CS_REQUIRE(pad_top < H && pad_bottom < H, "ingest: reflect padding must be smaller than the frame (np.pad 'reflect')");
  CS_REQUIRE((img == nullptr) == (x_nchw == nullptr && x_nhwc4 == nullptr), "ingest: image input and image outputs go together");
  CS_REQUIRE((lbl == nullptr) == (labels == nullptr), "ingest: label input and label output go together");
  CS_REQUIRE((mean == nullptr) == (stdv == nullptr), "ingest: mean and std go together");
  CS_REQUIRE(x_nhwc4 == nullptr || cs_aligned16(x_nhwc4), "ingest: NHWC-4 output must be 16-byte aligned");
  const int Ho = H + pad_top + pad_bottom;
  const long long n = (long long)B * Ho * W;
  hipLaunchKernelGGL(ingest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, lbl, B, H, W, lut, flips,
                     pad_top, Ho, mean, stdv, x_nchw, x_nhwc4, labels);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
