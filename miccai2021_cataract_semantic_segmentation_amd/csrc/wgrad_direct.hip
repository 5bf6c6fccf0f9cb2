// Backward-weight of the small-channel 3x3 / stride 1 / pad 1 convolutions of the HRNet branches (48 -> 48, 96 -> 96:
// 128 of the 316 weight-gradient launches of an OCRNet-HRNet-W48 step) as a DIRECT convolution.
//
// Why not the implicit GEMM (igemm.hip, layout TN): with M = Cout = 48 the GEMM re-reads dy once per 64-column N tile
// (7 x) and x once per filter tap (9 x) through L2 -> LDS: 800 MB of LDS fills for 100 MB of operands, 16 FLOP per
// filled byte; the kernel is bound by that traffic (54 TFLOP/s).  Here a block walks over strips of 16 consecutive output
// pixels of one image row; per strip it loads dy[16][Cout] and the input rows x[y-1 .. y+1][x0-1 .. x0+16][Cin] ONCE
// (halo included) and derives all filter taps from shifted windows of that LDS image:
//   dw[o][ky][kx][c] += sum_px dy[px][o] * xs[ky][px + kx][c]
// v_mfma_f32_16x16x4_f32, M = Cout (3 tiles of 16 per wave), N = (ky, kx, c / 16) tiles spread over the waves, k = pixel.
// Partial sums per pixel range go to slabs that reduce_slabs_kernel (igemm.hip) adds in a fixed order: deterministic.
#include "common.h"

namespace {

template <int CO, int CI, int NKY, int WM, int WN>
struct WgCfg {
  static constexpr int SA = CO + 4, SB = CI + 4;                 // padded LDS row strides: (4 * stride) mod 64 = 16 -> the four
                                                                 // k-groups of a wave read disjoint bank quarters
  static constexpr int NA = 16 * CO / 4, NB = NKY * 18 * CI / 4;  // 16-byte chunks per strip
  static constexpr int ASLOTS = (NA + 255) / 256, BSLOTS = (NB + 255) / 256;
  static constexpr int TM = CO / 16 / WM;                         // accumulator tiles per wave, M side
  static constexpr int NT = NKY * 3 * (CI / 16);                  // N tiles of the block
  static constexpr int TN = (NT + WN - 1) / WN;
  static constexpr int LDS_FLOATS = 2 * (16 * SA + NKY * 18 * SB);
};

template <int CO, int CI, int NKY, int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_direct_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int ldy, int B,
                                                              int H, int W, int strips_per_block, float* __restrict__ slabs, long long slab_stride) {
  using C = WgCfg<CO, CI, NKY, WM, WN>;
  __shared__ __attribute__((aligned(16))) float smem[C::LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int ky0 = blockIdx.y * NKY;
  const int SW = (W + 15) >> 4;
  const long long S = (long long)B * H * SW;
  const long long s_begin = (long long)blockIdx.x * strips_per_block;
  const long long s_end = min(s_begin + strips_per_block, S);

  // ---- per-thread load slots: the loop-invariant part of each 16-byte chunk's address -------------------------
  // A slots: chunk q = (pixel, 4 output channels) of dy; B slots: chunk q = (ky row, halo pixel, 4 input channels) of x
  int a_px[C::ASLOTS], a_off[C::ASLOTS], a_lds[C::ASLOTS];
  int b_dy[C::BSLOTS], b_px[C::BSLOTS], b_off[C::BSLOTS], b_lds[C::BSLOTS];
#pragma unroll
  for (int i = 0; i < C::ASLOTS; ++i) {
    const int q = tid + i * 256;
    const int px = q / (CO / 4), c4 = q % (CO / 4);
    a_px[i] = q < C::NA ? px : 1 << 20;             // out of range -> never valid
    a_off[i] = px * ldy + c4 * 4;
    a_lds[i] = q < C::NA ? px * C::SA + c4 * 4 : -1;
  }
#pragma unroll
  for (int i = 0; i < C::BSLOTS; ++i) {
    const int q = tid + i * 256;
    const int kyl = q / (18 * CI / 4), rem = q % (18 * CI / 4);
    const int px = rem / (CI / 4), c4 = rem % (CI / 4);
    b_dy[i] = q < C::NB ? kyl : 1 << 20;
    b_px[i] = px;
    b_off[i] = (kyl * W + px) * ldx + c4 * 4;
    b_lds[i] = q < C::NB ? 16 * C::SA + (kyl * 18 + px) * C::SB + c4 * 4 : -1;
  }

  f32x4 acc[C::TM][C::TN];
#pragma unroll
  for (int t = 0; t < C::TM; ++t)
#pragma unroll
    for (int u = 0; u < C::TN; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][u][r] = 0.f;

  // strip cursor (block-uniform): image b, row y, strip sx of 16 pixels
  int cs = (int)(s_begin % SW);
  int cy = (int)((s_begin / SW) % H);
  int cb = (int)(s_begin / ((long long)SW * H));
  f32x4 ra[C::ASLOTS], rb[C::BSLOTS];
  auto fetch = [&]() {   // loads the strip under the cursor, then advances the cursor
    const int x0 = cs * 16;
    const float* dyr = dy + ((long long)(cb * H + cy) * W + x0) * ldy;
    const float* xr = x + ((long long)(cb * H + cy + ky0 - 1) * W + x0 - 1) * ldx;   // may point before the row: guarded below
#pragma unroll
    for (int i = 0; i < C::ASLOTS; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (x0 + a_px[i] < W) v = *(const f32x4*)(dyr + a_off[i]);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < C::BSLOTS; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int yy = cy + ky0 - 1 + b_dy[i], xx = x0 - 1 + b_px[i];
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = *(const f32x4*)(xr + b_off[i]);
      rb[i] = v;
    }
    if (++cs == SW) {
      cs = 0;
      if (++cy == H) { cy = 0; ++cb; }
    }
  };
  auto stash = [&](int buf) {
    float* base = smem + buf * (C::LDS_FLOATS / 2);
#pragma unroll
    for (int i = 0; i < C::ASLOTS; ++i)
      if (a_lds[i] >= 0) *(f32x4*)(base + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < C::BSLOTS; ++i)
      if (b_lds[i] >= 0) *(f32x4*)(base + b_lds[i]) = rb[i];
  };
  auto compute = [&](int buf) {
    const float* sA = smem + buf * (C::LDS_FLOATS / 2);
    const float* sB = sA + 16 * C::SA;
    float a[C::TM][4];
#pragma unroll
    for (int t = 0; t < C::TM; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[t][e] = sA[(4 * g + e) * C::SA + (wm * C::TM + t) * 16 + i16];
#pragma unroll
    for (int u = 0; u < C::TN; ++u) {
      const int n = wn + u * WN;  // N tile of this wave: (ky_l, kx, c16), c16 fastest
      if (n < C::NT) {
        const int c16 = n % (CI / 16), kx = (n / (CI / 16)) % 3, kyl = n / (3 * (CI / 16));
        const float* bp = sB + (kyl * 18 + kx) * C::SB + c16 * 16 + i16;
        float bb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bb[e] = bp[(4 * g + e) * C::SB];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < C::TM; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][e], bb[e], acc[t][u], 0, 0, 0);
      }
    }
  };

  if (s_begin < s_end) {
    fetch();
    stash(0);
    __syncthreads();
    int cur = 0;
    for (long long s = s_begin; s < s_end; ++s) {
      const bool more = s + 1 < s_end;
      if (more) fetch();            // global loads in flight under the MFMAs
      compute(cur);
      if (more) stash(cur ^ 1);     // the other buffer was last read before the previous barrier
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- this block's partial sums: slab[split][o][ky][kx][c] ---------------------------------------------------
  float* out = slabs + (long long)blockIdx.x * slab_stride;
#pragma unroll
  for (int t = 0; t < C::TM; ++t)
#pragma unroll
    for (int u = 0; u < C::TN; ++u) {
      const int n = wn + u * WN;
      if (n < C::NT) {
        const int c16 = n % (CI / 16), kx = (n / (CI / 16)) % 3, kyl = n / (3 * (CI / 16));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = (wm * C::TM + t) * 16 + 4 * g + r;
          out[((m * 3 + ky0 + kyl) * 3 + kx) * CI + c16 * 16 + i16] = acc[t][u][r];
        }
      }
    }
}

int g_wgrad_direct = 1;
int g_direct_blocks = 512;

struct DirectPlan { int kind, splits, strips_per_block; };  // kind 0 = not applicable

DirectPlan direct_plan(const catseg_conv_desc* d) {
  DirectPlan p = {0, 0, 0};
  if (!g_wgrad_direct || d->stem4 || d->groups > 1) return p;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil != 1) return p;
  if (d->Cin == 48 && d->Cout == 48) p.kind = 1;
  else if (d->Cin == 96 && d->Cout == 96) p.kind = 2;
  else return p;
  const long long S = (long long)d->B * d->H * ((d->W + 15) / 16);
  if (S < 1024) { p.kind = 0; return p; }  // tiny maps: the GEMM path with its finer split is as good
  const int ky_blocks = p.kind == 1 ? 1 : 3;
  long long splits = g_direct_blocks / ky_blocks;   // default 512 blocks: ~2 resident blocks per CU
  if (splits > S / 8) splits = S / 8;
  p.strips_per_block = (int)((S + splits - 1) / splits);
  p.splits = (int)((S + p.strips_per_block - 1) / p.strips_per_block);
  return p;
}

}  // namespace

extern "C" int catseg_debug_set_wgrad_direct(int on) {
  if (on > 1) g_direct_blocks = on;   // tuning: values > 1 set the target number of blocks per launch
  g_wgrad_direct = on ? 1 : 0;
  return CATSEG_OK;
}

// used by catseg_conv2d_bwd_weight(_workspace) in igemm.hip
size_t wgrad_direct_workspace(const catseg_conv_desc* d) {
  const DirectPlan p = direct_plan(d);
  return p.kind ? (size_t)p.splits * d->Cout * 9 * d->Cin * 4 : 0;
}

// returns the number of slabs written to `workspace` (0 = not applicable: use the implicit GEMM); the caller reduces them
int wgrad_direct_launch(const catseg_conv_desc* d, const float* x, const float* dy, void* workspace, hipStream_t st) {
  const DirectPlan p = direct_plan(d);
  if (!p.kind) return 0;
  const long long wel = (long long)d->Cout * 9 * d->Cin;
  if (p.kind == 1)
    hipLaunchKernelGGL((wgrad_direct_kernel<48, 48, 3, 1, 4>), dim3(p.splits, 1), dim3(256), 0, st, x, d->ldx, dy, d->ldy, d->B, d->H, d->W,
                       p.strips_per_block, (float*)workspace, wel);
  else
    hipLaunchKernelGGL((wgrad_direct_kernel<96, 96, 1, 2, 2>), dim3(p.splits, 3), dim3(256), 0, st, x, d->ldx, dy, d->ldy, d->B, d->H, d->W,
                       p.strips_per_block, (float*)workspace, wel);
  return p.splits;
}
