// Backward-weight of the 3x3 / stride 1 / pad 1 convolutions of the HRNet trunk (Cin = Cout = 48, 96, 192, 384; reference
// models/HRNetv2.py:22-65, autograd of F.conv2d) as a DIRECT convolution in split precision:
//   dw[o][ky][kx][c] = sum_px dy[px][o] * x[px + (ky - 1, kx - 1)][c]
// A block owns a run of 4 x 16 pixel tiles; per tile it loads dy[64][COT] and the input rows x[rows + ky][18][NCI] as fp32 ONCE
// (buffer loads, rows / columns outside the image come back as zeros), splits them exactly into three bf16 planes in registers and
// keeps them in LDS pixel-major (channels contiguous, as in HBM); every filter tap is a shifted window of that image.
// v_mfma_f32_16x16x32_bf16 with k = pixel: both operands are k-strided in LDS, so the fragments (8 pixels of one channel) are read
// with the hardware transpose ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group).  Six products per block (hh hl lh hm mh mm),
// fp32 accumulation: the arithmetic of igemm_bf16x3.hip.  The accumulators (COT x taps x NCI per block) stay in registers over the
// whole run; per-block partial sums go to slabs that are added in a fixed order (deterministic).
//
// Bank conflicts: a transposed read serves 32 lanes per LDS cycle = 8 pixel rows x 32 bytes.  The k index of a K-step (32 pixels =
// 2 tile rows) is permuted so that those are 8 CONSECUTIVE pixels of one row (lane group g takes pixels 4g .. 4g+3 of row 0 as
// k = 0..3 and of row 1 as k = 4..7; both operands use the same permutation), and the pixel row stride is an odd multiple of 32 bytes:
// 8 consecutive rows then fall into 8 different 32-byte bank groups.
#include "common.h"

// This file compiles twice (as dconv3_b3.hip does): as it is -- three bf16 planes, six products -- and through dwgrad3_f16x2.hip (-DDW_H2):
// TWO fp16 planes, THREE products; both operands are scaled by 2^e from their producers' amax records while they are split in registers
// and the partial sums are scaled back when the slabs are written; entry point catseg_dwgrad3_f16x2.
#ifdef DW_H2
#define DW_NPL 2
#define DW_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#define dwgrad3_b3_kernel dwgrad3_h2_kernel
#define dwgrad3_reduce_kernel dwgrad3_h2_reduce_kernel
#else
#define DW_NPL 3
#define DW_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

namespace {

#ifdef DW_H2
typedef _Float16 dw_t;
__device__ __forceinline__ int dw_exponent(unsigned amax_bits) {      // igemm_f16x2.hip: amax * 2^e in [2^14, 2^15)
  const int ex = (int)((amax_bits >> 23) & 0xFF);
  if (ex == 0 || ex == 255) return 0;
  const int e = 14 - (ex - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
#else
typedef __bf16 dw_t;
#endif
typedef dw_t bf16x8 __attribute__((ext_vector_type(8)));     // eight 16-bit plane elements (fp16 in the DW_H2 build)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int wg_rowbytes(int c) { return ((c * 2 / 32) & 1) ? c * 2 : c * 2 + 32; }

template <int COT_, int NCI_, int NTY_, int WM_, int WN_>
struct WgCfg {
  static constexpr int COT = COT_, NCI = NCI_, NTY = NTY_, WM = WM_, WN = WN_;
  static constexpr int TH = 4, TW = 16;
  static constexpr int NW = WM * WN, NTHR = 64 * NW;
  static constexpr int MT = COT / 16 / WM;              // output-channel tiles per wave
  static constexpr int NTL = NTY * 3 * (NCI / 16);      // n tiles of a block: (ky_local, kx, ci tile), ci tile fastest
  static constexpr int NTW = (NTL + WN - 1) / WN;       // n tiles per wave: wn, wn + WN, ...
  static constexpr int XH = TH + NTY - 1, XW = TW + 2;  // staged input rows / columns
  static constexpr int NPX_X = XH * XW, NPX_D = TH * TW;
  static constexpr int ROWB_X = wg_rowbytes(NCI), ROWB_D = wg_rowbytes(COT);
  static constexpr int XPS = (NPX_X * ROWB_X + 255) / 256 * 256, DPS = (NPX_D * ROWB_D + 255) / 256 * 256;
  static constexpr int D0 = DW_NPL * XPS;
  static constexpr int LDS = DW_NPL * (XPS + DPS);
  static constexpr int NKS = TH * TW / 32;
  static constexpr int NGB_X = (NCI / 8 + 3) / 4, NGB_D = (COT / 8 + 3) / 4;
  static constexpr int NU_X = ((NPX_X + 15) / 16) * NGB_X, NU_D = (NPX_D / 16) * NGB_D;
  static constexpr int IPT = (NU_X + NU_D + NW - 1) / NW;
  static_assert(COT % (16 * WM) == 0 && NCI % 16 == 0 && (NTY == 1 || NTY == 3), "tiling");
};

struct WgArgs {
  const float* x;
  int ldx;
  const float* dy;
  int ldy;
  int B, H, W, C;
  int tiles_y, tiles_x;
  float* slabs;          // [splits][C][9][C]
  long long slab_stride;
  const unsigned* x_rec;   // DW_H2: the operands' amax records (common.h: cs_amax_read)
  const unsigned* dy_rec;
};

template <class G>
__global__ __launch_bounds__(G::NTHR, 2) void dwgrad3_b3_kernel(const WgArgs a) {
  __shared__ __attribute__((aligned(256))) unsigned char smem[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int i16 = lane & 15, kg = lane >> 4;

  // ---- variant of this block: (output-channel tile, input-channel chunk, filter row) ---------------------------------------------
  const int nco = a.C / G::COT, nci = a.C / G::NCI;
  int v = blockIdx.y;
  const int cot0 = (v % nco) * G::COT; v /= nco;
  const int ci0 = (v % nci) * G::NCI; v /= nci;
  const int ky0 = G::NTY == 3 ? 0 : v;             // first filter row of this block

  const int ntile = a.B * a.tiles_y * a.tiles_x;
  const int t_begin = (int)((long long)blockIdx.x * ntile / gridDim.x), t_end = (int)((long long)(blockIdx.x + 1) * ntile / gridDim.x);
#ifdef DW_H2
  const int ex_x = __builtin_amdgcn_readfirstlane(dw_exponent(cs_amax_read(a.x_rec)));
  const int ex_d = __builtin_amdgcn_readfirstlane(dw_exponent(cs_amax_read(a.dy_rec)));
  const float sc_x = __builtin_ldexpf(1.f, ex_x), sc_d = __builtin_ldexpf(1.f, ex_d);     // (|e| <= 100: normal floats; x * 2^e is exact)
  const bool one_step = ex_x + ex_d >= -120 && ex_x + ex_d <= 120;
  const float sc_out = __builtin_ldexpf(1.f, one_step ? -(ex_x + ex_d) : 0);
#endif

  // ---- staging items: unit u = wave + i NW; units [0, NU_X) = input rows, the rest = dy; lane = pixel + 16 (channel group) ----
  f32x4 pre[G::IPT][2];
  auto fetch = [&](int tile) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)b * a.H * a.W * a.ldx + ci0), (short)0,
                                                                        ((a.H * a.W - 1) * a.ldx + G::NCI) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (long long)b * a.H * a.W * a.ldy + cot0), (short)0,
                                                                        ((a.H * a.W - 1) * a.ldy + G::COT) * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < G::IPT; ++i) {
      const int u = wave + i * G::NW;
      int off;
      if (u < G::NU_X) {
        const int px = (u / G::NGB_X) * 16 + i16, g8 = (u % G::NGB_X) * 4 + kg;
        const int iy = y0 + px / G::XW + ky0 - 1, ix = x0 + px % G::XW - 1;     // rows outside the image: out of range by themselves
        const bool ok = (unsigned)ix < (unsigned)a.W && px < G::NPX_X && g8 < G::NCI / 8;
        off = ok ? ((iy * a.W + ix) * a.ldx + g8 * 8) * 4 : (int)0xFFFFFFE0;
        pre[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
        pre[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off + 16, 0, 0));
      } else {
        const int ud = u - G::NU_X;
        const int px = (ud / G::NGB_D) * 16 + i16, g8 = (ud % G::NGB_D) * 4 + kg;
        const int iy = y0 + px / G::TW, ix = x0 + px % G::TW;
        const bool ok = ix < a.W && ud < G::NU_D && g8 < G::COT / 8;
        off = ok ? ((iy * a.W + ix) * a.ldy + g8 * 8) * 4 : (int)0xFFFFFFE0;
        pre[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsd, off, 0, 0));
        pre[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsd, off + 16, 0, 0));
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < G::IPT; ++i) {
      const int u = wave + i * G::NW;
      bf16x8 h, m;
#ifdef DW_H2
      const float sc = u < G::NU_X ? sc_x : sc_d;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xs = pre[i][j >> 2][j & 3] * sc;
        const _Float16 hh = (_Float16)xs;
        h[j] = hh;
        m[j] = (_Float16)(xs - (float)hh);
      }
#else
      bf16x8 l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float vv = pre[i][j >> 2][j & 3];
        const __bf16 hh = (__bf16)vv;
        const float r1 = vv - (float)hh;
        const __bf16 mm = (__bf16)r1;
        h[j] = hh;
        m[j] = mm;
        l[j] = (__bf16)(r1 - (float)mm);
      }
#endif
      int dst, ps;
      bool live;
      if (u < G::NU_X) {
        const int px = (u / G::NGB_X) * 16 + i16, g8 = (u % G::NGB_X) * 4 + kg;
        live = px < G::NPX_X && g8 < G::NCI / 8;
        dst = px * G::ROWB_X + g8 * 16;
        ps = G::XPS;
      } else {
        const int ud = u - G::NU_X;
        const int px = (ud / G::NGB_D) * 16 + i16, g8 = (ud % G::NGB_D) * 4 + kg;
        live = ud < G::NU_D && g8 < G::COT / 8;
        dst = G::D0 + px * G::ROWB_D + g8 * 16;
        ps = G::DPS;
      }
      if (live) {
        *(bf16x8*)(smem + dst) = h;
        *(bf16x8*)(smem + dst + ps) = m;
#ifndef DW_H2
        *(bf16x8*)(smem + dst + 2 * ps) = l;
#endif
      }
    }
  };

  // ---- fragment addresses: lane = 16 g + 4 q + pp supplies (pixel row q of its group's block, columns 4 pp .. 4 pp + 3) ---------
  const int q = (lane >> 2) & 3, pp = lane & 3;
  const int a_base = G::D0 + (4 * kg + q) * G::ROWB_D + wm * G::MT * 32 + pp * 8;   // + ks * 32 ROWB_D + h * 16 ROWB_D + mt * 32 + p * DPS
  const int b_base = (4 * kg + q) * G::ROWB_X + pp * 8;                            // + ks * 2 XW ROWB_X + h * XW ROWB_X + tap / ci-tile offset
  int noff[G::NTW];   // n tile -> byte offset of its tap window / channel tile in the input image (wave uniform)
#pragma unroll
  for (int j = 0; j < G::NTW; ++j) {
    const int n = wn + j * G::WN;
    const int ct = n % (G::NCI / 16), kx = (n / (G::NCI / 16)) % 3, kyl = n / (3 * (G::NCI / 16));
    noff[j] = (kyl * G::XW + kx) * G::ROWB_X + ct * 32;
  }

  f32x4 acc[G::MT][G::NTW];
#pragma unroll
  for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
    for (int j = 0; j < G::NTW; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tread = [&](int addr) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + addr));
    return lo;
  };
  auto frag = [&](int addr, int hstride) {   // k = 0..3: the group's 4 pixels of tile row 2 ks, k = 4..7: of row 2 ks + 1
    const s16x4 lo = tread(addr), hi = tread(addr + hstride);
    const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v8);
  };

  if (t_begin < t_end) fetch(t_begin);
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; ++tile) {
    stash();                               // (the previous tile's fragment reads are behind the barrier that closed its last K-step)
    __syncthreads();
    if (tile + 1 < t_end) fetch(tile + 1);   // in flight under this tile's MFMAs
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks) {
      bf16x8 af[G::MT][DW_NPL];
#pragma unroll
      for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
        for (int p = 0; p < DW_NPL; ++p) af[mt][p] = frag(a_base + ks * 32 * G::ROWB_D + mt * 32 + p * G::DPS, 16 * G::ROWB_D);
#pragma unroll
      for (int j = 0; j < G::NTW; ++j) {
        if (wn + j * G::WN < G::NTL) {
          bf16x8 bf[DW_NPL];
#pragma unroll
          for (int p = 0; p < DW_NPL; ++p) bf[p] = frag(b_base + ks * 2 * G::XW * G::ROWB_X + noff[j] + p * G::XPS, G::XW * G::ROWB_X);
#pragma unroll
          for (int mt = 0; mt < G::MT; ++mt) {
            f32x4 c = acc[mt][j];
#if DW_NPL == 3
            c = DW_MFMA(af[mt][1], bf[1], c, 0, 0, 0);   // m m
            c = DW_MFMA(af[mt][0], bf[2], c, 0, 0, 0);   // h l
            c = DW_MFMA(af[mt][2], bf[0], c, 0, 0, 0);   // l h
#endif
            c = DW_MFMA(af[mt][0], bf[1], c, 0, 0, 0);   // h m   (DW_H2: plane 1 = l)
            c = DW_MFMA(af[mt][1], bf[0], c, 0, 0, 0);   // m h
            c = DW_MFMA(af[mt][0], bf[0], c, 0, 0, 0);   // h h
            acc[mt][j] = c;
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- this block's partial sums: slab[split][o][ky][kx][c] --------------------------------------------------------------------
  float* out = a.slabs + (long long)blockIdx.x * a.slab_stride;
#pragma unroll
  for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
    for (int j = 0; j < G::NTW; ++j) {
      const int n = wn + j * G::WN;
      if (n < G::NTL) {
        const int ct = n % (G::NCI / 16), kx = (n / (G::NCI / 16)) % 3, ky = ky0 + n / (3 * (G::NCI / 16));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = cot0 + (wm * G::MT + mt) * 16 + 4 * kg + r;
#ifdef DW_H2
          out[((long long)(o * 3 + ky) * 3 + kx) * a.C + ci0 + ct * 16 + i16] =
              one_step ? acc[mt][j][r] * sc_out : __builtin_ldexpf(__builtin_ldexpf(acc[mt][j][r], -ex_x), -ex_d);
#else
          out[((long long)(o * 3 + ky) * 3 + kx) * a.C + ci0 + ct * 16 + i16] = acc[mt][j][r];
#endif
        }
      }
    }
}

// dw = sum over the slabs, in a fixed order: 64 float4 columns (1 KB contiguous per slab row) x 4 slab lanes per block, each lane adds
// every 4th slab with four independent load chains, fixed-order combine over the lanes
__global__ __launch_bounds__(256) void dwgrad3_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, long long n4, int splits,
                                                             long long slab_stride4) {
  __shared__ f32x4 sh[4][64];
  const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long col = (long long)blockIdx.x * 64 + c;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (col < n4) {
    const f32x4* p = (const f32x4*)slabs + col;
    int k = sl;
    for (; k + 12 < splits; k += 16) {
      s0 += p[(long long)k * slab_stride4];
      s1 += p[(long long)(k + 4) * slab_stride4];
      s2 += p[(long long)(k + 8) * slab_stride4];
      s3 += p[(long long)(k + 12) * slab_stride4];
    }
    for (; k < splits; k += 4) s0 += p[(long long)k * slab_stride4];
  }
  sh[sl][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && col < n4) ((f32x4*)dw)[col] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
}

using Wg48 = WgCfg<48, 48, 3, 1, 4>;   // one block: all 48 x 432 accumulators (wave = 48 co x 7 of the 27 (tap, ci) tiles)
using Wg64 = WgCfg<64, 64, 1, 1, 4>;   // 64 x 64 (stage-1 bottlenecks): block = 64 co x (one filter row x 64 ci)
using Wg96 = WgCfg<96, 48, 1, 2, 2>;   // block = 96 co x (one filter row x 48 ci); variants over (co tile, ci chunk, filter row)

}  // namespace

// tuning hook (catseg_debug_set_dwgrad3_blocks): one variable for both builds of this source (external linkage, defined in the bf16x3 object)
#ifndef DW_H2
int catseg_g_wg_blocks = 512;
#else
extern int catseg_g_wg_blocks;
#endif
#define g_wg_blocks catseg_g_wg_blocks

namespace {

struct WgPlan { int kind, variants, splits; };

WgPlan wg_plan(int C, int B, int H, int W) {
  WgPlan p = {0, 0, 0};
  if (C == 48) { p.kind = 1; p.variants = 1; }
  else if (C == 64) { p.kind = 3; p.variants = 3; }
  else if (C == 96 || C == 192 || C == 384) { p.kind = 2; p.variants = (C / 96) * (C / 48) * 3; }
  else return p;
  const int ntile = B * ((H + 3) / 4) * ((W + 15) / 16);
  int s = g_wg_blocks / p.variants;
  if (s < 1) s = 1;
  if (s > ntile) s = ntile;
  p.splits = s;
  return p;
}

}  // namespace

namespace {
int wg_run(int B, int H, int W, int C, const float* x, int ldx, const float* dy, int ldy, float* dw, void* workspace, size_t workspace_bytes,
           const void* x_rec, const void* dy_rec, catseg_stream_t stream);
}

#ifdef DW_H2
// catseg_dwgrad3 on two fp16 planes: x_record / dy_record = the operands' amax records (include/catseg.h: catseg_dconv3_f16x2);
// workspace as catseg_dwgrad3_workspace
extern "C" int catseg_dwgrad3_f16x2(int B, int H, int W, int C, const float* x, int ldx, const void* x_record, const float* dy, int ldy,
                                    const void* dy_record, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(x_record && dy_record, "dwgrad3 f16x2: amax records missing");
  return wg_run(B, H, W, C, x, ldx, dy, ldy, dw, workspace, workspace_bytes, x_record, dy_record, stream);
}
#else
extern "C" int catseg_debug_set_dwgrad3_blocks(int blocks) {
  g_wg_blocks = blocks > 0 ? blocks : 512;
  return CATSEG_OK;
}

extern "C" int catseg_dwgrad3_supported(int C) { return C == 48 || C == 64 || C == 96 || C == 192 || C == 384; }

extern "C" size_t catseg_dwgrad3_workspace(int B, int H, int W, int C) {
  const WgPlan p = wg_plan(C, B, H, W);
  return p.kind ? (size_t)p.splits * C * 9 * C * 4 : 0;
}

extern "C" int catseg_dwgrad3(int B, int H, int W, int C, const float* x, int ldx, const float* dy, int ldy, float* dw, void* workspace,
                              size_t workspace_bytes, catseg_stream_t stream) {
  return wg_run(B, H, W, C, x, ldx, dy, ldy, dw, workspace, workspace_bytes, nullptr, nullptr, stream);
}
#endif

namespace {
int wg_run(int B, int H, int W, int C, const float* x, int ldx, const float* dy, int ldy, float* dw, void* workspace, size_t workspace_bytes,
           const void* x_rec, const void* dy_rec, catseg_stream_t stream) {
  const WgPlan p = wg_plan(C, B, H, W);
  CS_REQUIRE(p.kind, "dwgrad3: unsupported channel count %d", C);
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && x && dy && dw && workspace, "dwgrad3: bad args");
  CS_REQUIRE(ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && cs_aligned16(x) && cs_aligned16(dy) && cs_aligned16(dw) &&
                 cs_aligned16(workspace), "dwgrad3: alignment / row strides");
  CS_REQUIRE((long long)H * W * (long long)(ldx > ldy ? ldx : ldy) < (1LL << 29), "dwgrad3: image too large for 32-bit byte offsets");
  const long long wel = (long long)C * 9 * C;
  if (workspace_bytes < (size_t)p.splits * wel * 4) {
    catseg_set_error("dwgrad3: workspace %zu < %zu", workspace_bytes, (size_t)p.splits * wel * 4);
    return CATSEG_EWORKSPACE;
  }
  WgArgs a;
  a.x = x; a.ldx = ldx; a.dy = dy; a.ldy = ldy;
  a.B = B; a.H = H; a.W = W; a.C = C;
  a.tiles_y = (H + 3) / 4;
  a.tiles_x = (W + 15) / 16;
  a.slabs = (float*)workspace;
  a.slab_stride = wel;
  a.x_rec = (const unsigned*)x_rec; a.dy_rec = (const unsigned*)dy_rec;
  hipStream_t st = (hipStream_t)stream;
  if (p.kind == 1) hipLaunchKernelGGL((dwgrad3_b3_kernel<Wg48>), dim3(p.splits, p.variants), dim3(Wg48::NTHR), 0, st, a);
  else if (p.kind == 3) hipLaunchKernelGGL((dwgrad3_b3_kernel<Wg64>), dim3(p.splits, p.variants), dim3(Wg64::NTHR), 0, st, a);
  else hipLaunchKernelGGL((dwgrad3_b3_kernel<Wg96>), dim3(p.splits, p.variants), dim3(Wg96::NTHR), 0, st, a);
  const long long n4 = wel / 4;
  hipLaunchKernelGGL(dwgrad3_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, (const float*)workspace, dw, n4, p.splits, n4);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
}  // namespace
