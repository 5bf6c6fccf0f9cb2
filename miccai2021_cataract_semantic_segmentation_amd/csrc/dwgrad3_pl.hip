// Backward-weight of the 3x3 / stride 1 / pad 1 convolutions of the HRNet trunk on PRODUCER-WRITTEN fp16 x 2 operand planes (round 4):
//   dw[o][ky][kx][c] = sum_px dy[px][o] * x[px + (ky - 1, kx - 1)][c]              (autograd of F.conv2d, models/HRNetv2.py:22-65)
// dwgrad3_f16x2.hip (= dwgrad3_b3.hip with DW_H2) loads BOTH operands as fp32, scales and splits them into two fp16 planes in registers
// and stores them to LDS, in the waves that then issue the MFMAs; fabric traffic 216 MB per layer against 100 MB of operand bytes
// (profiles/r03_pmc_traffic_ocrnet_hrnet48.json).  Here both operands already exist as planes (csrc/planes.h: x from the forward pass'
// producer, dy from the BatchNorm backward), 4 bytes per element like the fp32 tensors they replace, and a tile's images stream into LDS
// by LDS-DMA (buffer_load ... lds, 16 bytes per lane = one (pixel, channel group)) straight into the pixel-major rows the transposed
// fragment reads (ds_read_b64_tr_b16) want: no VALU work, no register staging, no ds_write.  Two image buffers: the DMA of tile t + 1 is
// issued before the MFMAs of tile t and awaited behind them, ONE barrier per tile.
// Tiling, fragment addressing, product order (hl, lh, hh), slabs + fixed-order reduction: exactly dwgrad3_f16x2's -- the results are
// bit-identical to it for equal exponents (tests/test_dwgrad3_pl_gpu.py).
#include "planes.h"

extern int catseg_g_wg_blocks;      // csrc/dwgrad3_b3.hip (catseg_debug_set_dwgrad3_blocks): the 48-channel form (two blocks per CU)
// blocks of the 96+ channel form: THREE fit a CU (46 KB of LDS, four waves).  With 768 blocks a launch uses the third slot and hides its LDS-DMA
// round trips: standalone, reduction included, 61.7 -> 52.9 us (96 channels), 58.9 -> 49.6 (192), 67.5 -> 50.1 (384); 1024 is slower again.
// In the step the kernels of sibling streams fill that slot anyway (118.5 ms either way) and 768 blocks mean 1.5 x the slabs (PMC: 99 -> 132 MB
// of fabric traffic per layer): the default stays 512.
int catseg_g_wp96_blocks = 512;
extern "C" int catseg_debug_set_dwgrad3_pl_blocks(int blocks) {
  catseg_g_wp96_blocks = blocks > 0 ? blocks : 512;
  return CATSEG_OK;
}

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int wp_rowbytes(int c) { return ((c * 2 / 32) & 1) ? c * 2 : c * 2 + 32; }   // an odd multiple of 32 bytes (see dwgrad3_b3.hip)

template <int COT_, int NCI_, int NTY_, int WM_, int WN_, int TH_, int NBUF_>
struct WpCfg {
  static constexpr int NBUF = NBUF_;                    // image buffers: the LDS-DMA of tile t + NBUF - 1 is issued in front of tile t's MFMAs
  static constexpr int COT = COT_, NCI = NCI_, NTY = NTY_, WM = WM_, WN = WN_, TH = TH_, TW = 16;
  static constexpr int NW = WM * WN, NTHR = 64 * NW;
  static_assert(NW == 4 && (TH == 2 || TH == 4), "four waves; tiles of 2 or 4 rows");
  static constexpr int MT = COT / 16 / WM;              // output-channel tiles per wave
  static constexpr int NTL = NTY * 3 * (NCI / 16);      // n tiles of a block: (ky_local, kx, ci tile), ci tile fastest
  static constexpr int NTW = (NTL + WN - 1) / WN;       // n tiles per wave: wn, wn + WN, ...
  static constexpr int XH = TH + NTY - 1, XW = TW + 2;  // staged input rows / columns
  static constexpr int NPX_X = XH * XW, NPX_D = TH * TW;
  static constexpr int ROWB_X = wp_rowbytes(NCI), ROWB_D = wp_rowbytes(COT);
  static constexpr int SX = ROWB_X / 16, SD = ROWB_D / 16;            // 16-byte slots per pixel row (channel groups + padding)
  static constexpr int XPS = (NPX_X * ROWB_X + 255) / 256 * 256, DPS = (NPX_D * ROWB_D + 255) / 256 * 256;
  static constexpr int D0 = 2 * XPS, BUF = 2 * (XPS + DPS);
  static constexpr int LDS = NBUF * BUF;
  static constexpr int NKS = TH * TW / 32;
  static constexpr int NJX = (NPX_X * SX + 63) / 64, NJD = (NPX_D * SD + 63) / 64;     // LDS-DMA instructions per plane
  static constexpr int MX = (NJX + 3) / 4, MD = (NJD + 3) / 4;                          // ... per wave
  static_assert(COT % (16 * WM) == 0 && NCI % 16 == 0 && (NTY == 1 || NTY == 3) && LDS <= 80 * 1024, "tiling");
};

struct WpArgs {
  const unsigned char* xp;      // planes of x  [2][C / 8][P][8]
  const unsigned char* dp;      // planes of dy
  unsigned P16;
  const int* x_rec;             // word CS_REC_EXP = exponent of the planes
  const int* d_rec;
  int B, H, W, C;
  int tiles_y, tiles_x, splits;
  float* slabs;                 // [splits][C][9][C]
  long long slab_stride;
};

template <class G>
__global__ __launch_bounds__(G::NTHR, 2) void dwgrad3_pl_kernel(const WpArgs a) {
  __shared__ __attribute__((aligned(256))) unsigned char smem[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int i16 = lane & 15, kg = lane >> 4;

  // ---- (split, variant) of this block.  Block ids that share an XCD (id % 8) walk the VARIANTS of one split one after the other: the
  //      (output-channel tile, input-channel chunk, filter row) variants of a split read the same pixel tiles of x and dy, which then come
  //      from that XCD's L2 instead of crossing the fabric once per variant (216 MB per layer against 100 MB of operands in round 3) ----
  const int nco = a.C / G::COT, nci = a.C / G::NCI;
  const int nvar = nco * nci * (G::NTY == 3 ? 1 : 3);
  const int split = (int)(blockIdx.x >> 3) / nvar * 8 + (int)(blockIdx.x & 7);
  if (split >= a.splits) return;
  int v = (int)(blockIdx.x >> 3) % nvar;
  const int cot0 = (v % nco) * G::COT; v /= nco;
  const int ci0 = (v % nci) * G::NCI; v /= nci;
  const int ky0 = G::NTY == 3 ? 0 : v;             // first filter row of this block
  const int NG = a.C / 8;

  const int ntile = a.B * a.tiles_y * a.tiles_x;
  const int t_begin = (int)((long long)split * ntile / a.splits), t_end = (int)((long long)(split + 1) * ntile / a.splits);
  const int ex_x = __builtin_amdgcn_readfirstlane(a.x_rec[CS_REC_EXP]);
  const int ex_d = __builtin_amdgcn_readfirstlane(a.d_rec[CS_REC_EXP]);
  const bool one_step = ex_x + ex_d >= -120 && ex_x + ex_d <= 120;
  const float sc_out = __builtin_ldexpf(1.f, one_step ? -(ex_x + ex_d) : 0);

  // ---- LDS-DMA items of this wave: instruction j = wave + 4 m of an image covers its 16-byte slots 64 j .. 64 j + 63; slot s = pixel
  //      s / S, channel group s % S (groups past the operand's channel chunk are the row padding: out of range -> zeros) -----------------
  int x_rel[G::MX], x_r[G::MX], x_c[G::MX], d_rel[G::MD], d_r[G::MD], d_c[G::MD];
  unsigned x_go[G::MX], d_go[G::MD];
#pragma unroll
  for (int m = 0; m < G::MX; ++m) {
    const int s = (wave + 4 * m) * 64 + lane, px = s / G::SX, g = s - px * G::SX;
    const bool live = px < G::NPX_X && g < G::NCI / 8;
    x_r[m] = live ? px / G::XW + ky0 - 1 : -(1 << 20);
    x_c[m] = px % G::XW - 1;
    x_rel[m] = ((px / G::XW + ky0 - 1) * a.W + (px % G::XW - 1)) * 16;
    x_go[m] = (unsigned)g * a.P16;
  }
#pragma unroll
  for (int m = 0; m < G::MD; ++m) {
    const int s = (wave + 4 * m) * 64 + lane, px = s / G::SD, g = s - px * G::SD;
    const bool live = px < G::NPX_D && g < G::COT / 8;
    d_r[m] = live ? px / G::TW : -(1 << 20);
    d_c[m] = px % G::TW;
    d_rel[m] = ((px / G::TW) * a.W + px % G::TW) * 16;
    d_go[m] = (unsigned)g * a.P16;
  }
  const unsigned plane_bytes = (unsigned)NG * a.P16;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.xp, (short)0, (int)(2u * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dp, (short)0, (int)(2u * plane_bytes), 0x00020000);
  auto dma = [&](int tile, int buf) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const int org = (y0 * a.W + x0) * 16;
    const unsigned img = (unsigned)(b * a.H * a.W) * 16u;
    unsigned char* base = smem + buf * G::BUF;
#pragma unroll
    for (int m = 0; m < G::MX; ++m) {
      const int j = wave + 4 * m;
      if (j < G::NJX) {
        const bool ok = (unsigned)(y0 + x_r[m]) < (unsigned)a.H && (unsigned)(x0 + x_c[m]) < (unsigned)a.W;
        const unsigned voff = ok ? (unsigned)(org + x_rel[m]) + x_go[m] : 0xFFFFFFF0u;
#pragma unroll
        for (int p = 0; p < 2; ++p)
          if (j * 64 + lane < G::NPX_X * G::SX)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(base + p * G::XPS + j * 1024), 16, voff,
                                                     (unsigned)(p * NG + ci0 / 8) * a.P16 + img, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < G::MD; ++m) {
      const int j = wave + 4 * m;
      if (j < G::NJD) {
        const bool ok = (unsigned)(y0 + d_r[m]) < (unsigned)a.H && (unsigned)(x0 + d_c[m]) < (unsigned)a.W;
        const unsigned voff = ok ? (unsigned)(org + d_rel[m]) + d_go[m] : 0xFFFFFFF0u;
#pragma unroll
        for (int p = 0; p < 2; ++p)
          if (j * 64 + lane < G::NPX_D * G::SD)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (__attribute__((address_space(3))) void*)(base + G::D0 + p * G::DPS + j * 1024), 16, voff,
                                                     (unsigned)(p * NG + cot0 / 8) * a.P16 + img, 0, 0);
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

  auto tread = [&](int addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + addr)); };
  auto frag = [&](int addr, int hstride) {   // k = 0..3: the group's 4 pixels of tile row 2 ks, k = 4..7: of row 2 ks + 1
#ifdef WP_NO_DSREAD
    const s16x4 lo = {(short)addr, (short)lane, 1, 2}, hi = {(short)hstride, 3, (short)tid, 4};
#else
    const s16x4 lo = tread(addr), hi = tread(addr + hstride);
#endif
    const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(h8, v8);
  };

  // LDS-DMA instructions this wave issues per tile (wave-uniform): the counted wait behind a tile's MFMAs leaves the NBUF - 2 youngest tiles in flight
  int n_items = 0;
#pragma unroll
  for (int m = 0; m < G::MX; ++m) n_items += (wave + 4 * m < G::NJX) ? 2 : 0;
#pragma unroll
  for (int m = 0; m < G::MD; ++m) n_items += (wave + 4 * m < G::NJD) ? 2 : 0;
  auto wait_vm = [&](int n) {
    switch (n) {
      case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
      case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
      case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
      case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
      case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
      case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
      case 12: __builtin_amdgcn_s_waitcnt(0x0F7C); break;
      default: __builtin_amdgcn_s_waitcnt(0x0F70); break;      // (more than 12 in flight per tile: wait for everything)
    }
  };
#pragma unroll
  for (int i = 0; i < G::NBUF - 1; ++i)
    if (t_begin + i < t_end) dma(t_begin + i, i);
  if (t_begin + G::NBUF - 2 < t_end) wait_vm((G::NBUF - 2) * n_items);   // the first tile has landed (the prologue's later ones may still fly)
  else __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  int buf = 0, nbuf = G::NBUF - 1;         // buffer of this tile / of the tile fetched in front of it
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; ++tile) {
    // (its buffer held tile - 1, whose last fragment reads are behind the barrier that closed that tile)
#ifndef WP_NO_DMA       // (differential-timing builds, tools/ab_wp.sh: wrong results, the time difference is the ingredient's cost)
    if (tile + G::NBUF - 1 < t_end) dma(tile + G::NBUF - 1, nbuf);
#endif
    const int bo = buf * G::BUF;
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks) {
      h8 af[G::MT][2];
#pragma unroll
      for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[mt][p] = frag(bo + a_base + ks * 32 * G::ROWB_D + mt * 32 + p * G::DPS, 16 * G::ROWB_D);
#pragma unroll
      for (int j = 0; j < G::NTW; ++j) {
        if (wn + j * G::WN < G::NTL) {
          h8 bf[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) bf[p] = frag(bo + b_base + ks * 2 * G::XW * G::ROWB_X + noff[j] + p * G::XPS, G::XW * G::ROWB_X);
#pragma unroll
          for (int mt = 0; mt < G::MT; ++mt) {
            f32x4 c = acc[mt][j];
#ifdef WP_NO_MFMA
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][0] + af[mt][1], bf[1] + bf[0], c, 0, 0, 0);
#else
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][0], bf[1], c, 0, 0, 0);   // h l
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][1], bf[0], c, 0, 0, 0);   // l h
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][0], bf[0], c, 0, 0, 0);   // h h
#endif
            acc[mt][j] = c;
          }
        }
      }
    }
    // this wave's pieces of the NEXT tile have landed (the NBUF - 2 tiles behind it may stay in flight; at the end of the run fewer were
    // issued: wait for everything) ...
    if (tile + G::NBUF - 1 < t_end) wait_vm((G::NBUF - 2) * n_items);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                         // ... everyone's have, and every fragment read of this tile has returned
    buf = buf + 1 == G::NBUF ? 0 : buf + 1;
    nbuf = nbuf + 1 == G::NBUF ? 0 : nbuf + 1;
  }

  // ---- this block's partial sums: slab[split][o][ky][kx][c] --------------------------------------------------------------------
  float* out = a.slabs + (long long)split * a.slab_stride;
#ifdef WP_NO_STORE
  if (a.splits > 0) {
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
      for (int j = 0; j < G::NTW; ++j) t += acc[mt][j][0] + acc[mt][j][1] + acc[mt][j][2] + acc[mt][j][3];
    if (t == 1.2345e-30f) out[0] = t;
    return;
  }
#endif
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
          out[((long long)(o * 3 + ky) * 3 + kx) * a.C + ci0 + ct * 16 + i16] =
              one_step ? acc[mt][j][r] * sc_out : __builtin_ldexpf(__builtin_ldexpf(acc[mt][j][r], -ex_x), -ex_d);
        }
      }
    }
}

// dw = sum over the slabs, in a fixed order (dwgrad3_b3.hip: dwgrad3_reduce_kernel)
__global__ __launch_bounds__(256) void dwgrad3_pl_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, long long n4, int splits,
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

using Wp48 = WpCfg<48, 48, 3, 1, 4, 4, 2>;   // one block: all 48 x 432 accumulators, tiles 4 x 16, two image buffers of 33 KB
#ifdef WP_ALT96
using Wp96 = WpCfg<48, 48, 1, 1, 4, 4, 2>;   // A/B: 48 co x (one filter row x 48 ci), tiles 4 x 16
#elif defined(WP_ROWS3)
// A/B: block = 96 co x (ALL THREE filter rows x 48 ci): 42 accumulators per wave (224 registers), a tile is 126 MFMAs per wave and dy is
// fetched by 2 variants of a split instead of 6.  Standalone it wins (reduction included: 61.6 -> 59.0 us at 96 channels, 59.8 -> 58.3 at 192,
// 66.7 -> 56.4 at 384; the kernel alone 61.7 -> 55.2 us in the step), but its slabs are 3 x (85 MB per layer; reduction 8.6 -> 21 us), and the
// whole step LOSES 1.7 - 2 ms with it (tools/ab_lib_bench.sh, three alternating rounds on one box: 118.1 - 118.6 against 119.8 - 121.9 ms).
using Wp96 = WpCfg<96, 48, 3, 2, 2, 2, 2>;
#else
// block = 96 co x (one filter row x 48 ci), tiles 2 x 16, two image buffers of 23 KB: THREE blocks fit a CU (catseg_debug_dwgrad3_pl_occupancy),
// which is what hides the LDS-DMA round trip behind a 45-MFMA tile.  A third image buffer costs the third block: 60 -> 92 us (PMC: the same
// wave cycles over 1.5 x the wall time), not adopted.  Differential builds (tools/ab_wp.sh) price its LDS-DMA at 14 of 53 us
// (MFMAs alone 28, fragment reads 5, slab store 3).
using Wp96 = WpCfg<96, 48, 1, 2, 2, 2, 2>;
#endif

struct WpPlan { int kind, variants, splits, TH; };

WpPlan wp_plan(int C, int B, int H, int W) {
  WpPlan p = {0, 0, 0, 0};
  if (C == 48) { p.kind = 1; p.variants = 1; p.TH = Wp48::TH; }
  else if (C == 96 || C == 192 || C == 384) { p.kind = 2; p.variants = (C / Wp96::COT) * (C / Wp96::NCI) * (Wp96::NTY == 3 ? 1 : 3); p.TH = Wp96::TH; }
  else return p;
  const int ntile = B * ((H + p.TH - 1) / p.TH) * ((W + 15) / 16);
  int s = (p.kind == 2 ? catseg_g_wp96_blocks : catseg_g_wg_blocks) / p.variants;
  if (s < 1) s = 1;
  if (s > ntile) s = ntile;
  p.splits = s;
  return p;
}

}  // namespace

extern "C" int catseg_dwgrad3_pl_supported(int C) { return C == 48 || C == 96 || C == 192 || C == 384; }

// blocks of the backward-weight kernel the runtime places on one CU (two by design: four waves per block, two waves per SIMD)
extern "C" int catseg_debug_dwgrad3_pl_occupancy(int C) {
  int n = -1;
  hipError_t e = C == 48 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dwgrad3_pl_kernel<Wp48>, Wp48::NTHR, 0)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dwgrad3_pl_kernel<Wp96>, Wp96::NTHR, 0);
  return e == hipSuccess ? n : -1;
}

extern "C" size_t catseg_dwgrad3_pl_workspace(int B, int H, int W, int C) {
  const WpPlan p = wp_plan(C, B, H, W);
  return p.kind ? (size_t)p.splits * C * 9 * C * 4 : 0;
}

// catseg_dwgrad3_f16x2 on producer-written planes of x and dy (csrc/planes.h; the records' word 1 = the planes' exponents)
extern "C" int catseg_dwgrad3_pl(int B, int H, int W, int C, const void* x_planes, const void* x_record, const void* dy_planes,
                                 const void* dy_record, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  const WpPlan p = wp_plan(C, B, H, W);
  CS_REQUIRE(p.kind, "dwgrad3 (planes): unsupported channel count %d", C);
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && x_planes && dy_planes && x_record && dy_record && dw && workspace, "dwgrad3 (planes): bad args");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(dy_planes) && cs_aligned16(dw) && cs_aligned16(workspace), "dwgrad3 (planes): alignment");
  const long long P = (long long)B * H * W;
  CS_REQUIRE(4 * P * C < (1LL << 32), "dwgrad3 (planes): tensor too large for 32-bit offsets");
  const long long wel = (long long)C * 9 * C;
  if (workspace_bytes < (size_t)p.splits * wel * 4) {
    catseg_set_error("dwgrad3 (planes): workspace %zu < %zu", workspace_bytes, (size_t)p.splits * wel * 4);
    return CATSEG_EWORKSPACE;
  }
  WpArgs a;
  a.xp = (const unsigned char*)x_planes; a.dp = (const unsigned char*)dy_planes; a.P16 = (unsigned)(P * 16);
  a.x_rec = (const int*)x_record; a.d_rec = (const int*)dy_record;
  a.B = B; a.H = H; a.W = W; a.C = C;
  a.tiles_y = (H + p.TH - 1) / p.TH;
  a.tiles_x = (W + 15) / 16;
  a.slabs = (float*)workspace;
  a.slab_stride = wel;
  a.splits = p.splits;
  hipStream_t st = (hipStream_t)stream;
  const int grid = (p.splits + 7) / 8 * 8 * p.variants;
  if (p.kind == 1) hipLaunchKernelGGL((dwgrad3_pl_kernel<Wp48>), dim3(grid), dim3(Wp48::NTHR), 0, st, a);
  else hipLaunchKernelGGL((dwgrad3_pl_kernel<Wp96>), dim3(grid), dim3(Wp96::NTHR), 0, st, a);
  const long long n4 = wel / 4;
  hipLaunchKernelGGL(dwgrad3_pl_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, (const float*)workspace, dw, n4, p.splits, n4);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
