// Implicit-GEMM convolution / batched GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains, 157 TFLOP/s dense peak).
//
// One kernel template, three operand layouts:
//   L_NT  conv forward / GEMM "NT":  A rows = output pixels (K contiguous), B rows = out-channels (K contiguous)
//   L_NN  conv backward-data / "NN": A rows = pixels (K contiguous),        B = [k][n] (n contiguous)
//   L_TN  conv backward-weight/"TN": A = [r][m], B = [r][n], reduction over pixel rows r (split over blocks)
//
// Data path: global --(global_load_lds_dwordx4, 16 B per lane, no VGPR round trip)--> LDS
// (double buffered, one K-step of 16 in flight behind a counted vmcnt) --> ds_read_b128 /
// ds_read_b32 fragments --> MFMA.  The im2col matrix is never materialised: each lane
// computes the source address of its 16-byte chunk (out-of-image taps point at a zero page).
//
// LDS images (per K-step of BK = 16 floats):
//   K-contiguous operand: [rows][16] floats, 64-B rows; the four 16-B chunks of a row are stored
//     at position (chunk ^ ((row >> 2) & 3)) so that the 16 lanes of a ds_read_b128 group hit 16
//     distinct 4-bank slots (the swizzle is applied to the per-lane SOURCE address, the LDS
//     destination of a global_load_lds is lane-linear).
//   K-major operand: [16][cols] floats, read with conflict-free ds_read_b32 (lane = column).
// MFMA k-mapping: lane half h = lane >> 5 feeds k = 8*jj + 4*h + ii of the K-step to MFMA (jj, ii);
// A and B use the same mapping, the sum over k is order-insensitive up to fp32 rounding.
#include "common.h"

// LDS ring depth of the 128x128 and larger tiles.  A/B on one MI355X (tools/bench_conv.py, same process order):
// 3 slots / one barrier per K-step = 124 TFLOP/s forward, 2 slots / two barriers but 4 blocks per CU = 128.
#ifndef CATSEG_NBUF_NARROW
#define CATSEG_NBUF_NARROW 2
#endif
#ifndef CATSEG_NBUF_BIG
#define CATSEG_NBUF_BIG 2
#endif

namespace {

__device__ __attribute__((aligned(256))) float g_zero_page[64];

enum { L_NT = 0, L_NN = 1, L_TN = 2 };

struct Geo {
  const float* base;
  int mode;  // 0 plain rows, 1 conv-forward gather, 2 conv-backward-data gather, 3 stem (chunk = pixel)
  int rows;  // valid row indices [0, rows)
  int ld;    // floats per source pixel / row
  int H, W;  // spatial dims of the source tensor
  int Ho, Wo;  // spatial dims the row index decodes over
  int kw, stride, pad, dil;
};

struct IgemmArgs {
  Geo g;               // gathered operand (A for L_NT / L_NN, B for L_TN)
  const float* other;  // L_NT: B[n][ldo]  L_NN: B[k][ldo]  L_TN: A[r][ldo]
  float* C;
  const float* bias;
  int bias_bs;  // bias elements per batch item (grouped conv: Cout / groups)
  int M, N, ldc, ldo, tap_stride;
  int taps, Cred, Cred_b;
  int tilesM, tilesN;
  int zero_to, accumulate;
  long long g_bs, o_bs, c_bs;
  // L_TN only
  int splits, rows_per_split;
  long long c_split_stride;
  int c_tap_stride;
  int tap_cin;  // L_TN, FAST: > 0 = the N dimension is taps x tap_cin (tap of a column = col / tap_cin), grid.y = 1
  // FAST gather of L_NT / L_NN: the taps form an nky x nkx grid (i, j); source pixel of (row (b, y, x), tap (i, j)) =
  //   (y * row_s + row_o + dy0 + i * ddy, x * row_s + row_o + dx0 + j * ddx); its weights sit at filter tap
  //   (wy0 + i * wys) * kw + (wx0 + j * wxs).  All affine: no table loads in the K loop (scalar loads share
  //   lgkmcnt with the LDS reads and would stall them).  row_s > 0 marks the parameters as valid.
  int row_s, row_o;
  int nky, nkx, dy0, ddy, dx0, ddx, wy0, wys, wx0, wxs;
  // output row remap (strided backward-data, one launch per input-pixel parity class):
  //   GEMM row (b, a, c) over (g.Ho, g.Wo) -> pixel ((b * out_H + a * out_s + out_py) * out_W + c * out_s + out_px)
  int remap, out_s, out_py, out_px, out_H, out_W;
  int step_b, step_qy, step_rx;  // L_TN FAST: 16 rows ahead = step_b images + step_qy image rows + step_rx pixels (no division in the K loop)
  // fused inference epilogue (eval-mode BatchNorm folded into w / bias on the host side): v = act(v + residual)
  const float* residual;
  int ldr, relu;
  const float* zero;  // 256-byte zero page (kernel argument: no GOT load / lgkmcnt wait inside the K loop)
  float* bn_part;     // L_NT: per (M-tile, channel) BatchNorm partials [tilesM][3][N] (K, s1, s2), or nullptr
};

__device__ __forceinline__ void glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void decode_row(const Geo& g, int r, int& b, int& y, int& x) {
  const int hw = g.Ho * g.Wo;
  b = r / hw;
  const int rem = r - b * hw;
  y = rem / g.Wo;
  x = rem - y * g.Wo;
}

// Source address of one 16-byte chunk of the gathered operand (or the zero page).
// chunk = index of the 4-float chunk inside the tap's reduction range; cvalid = chunk in range.
__device__ __forceinline__ const float* gather_ptr(const Geo& g, const float* base, bool rvalid, int r, int b,
                                                   int y, int x, int ky, int kx, int chunk, bool cvalid) {
  long long pix;
  bool ok = rvalid && cvalid;
  if (g.mode == 0) {
    pix = r;
  } else if (g.mode == 1) {
    const int iy = y * g.stride - g.pad + ky * g.dil;
    const int ix = x * g.stride - g.pad + kx * g.dil;
    ok = ok && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
    pix = ((long long)b * g.H + iy) * g.W + ix;
  } else if (g.mode == 2) {
    const int ty = y + g.pad - ky * g.dil;
    const int tx = x + g.pad - kx * g.dil;
    const int oy = ty / g.stride, ox = tx / g.stride;
    ok = ok && ty >= 0 && tx >= 0 && oy * g.stride == ty && ox * g.stride == tx && oy < g.H && ox < g.W;
    pix = ((long long)b * g.H + oy) * g.W + ox;
  } else {  // stem: tap = ky only, the 8 chunks of the K range are 8 consecutive pixels of 4 channels
    const int iy = y * g.stride - g.pad + ky * g.dil;
    const int ix = x * g.stride - g.pad + chunk;
    ok = ok && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
    pix = ((long long)b * g.H + iy) * g.W + ix;
    chunk = 0;
  }
  return ok ? base + pix * g.ld + chunk * 4 : g_zero_page;
}

// FAST: the gathered operand is a plain / stride-any forward conv gather (mode 1) or a stride-1
// backward-data gather (mode 2) with <= 32 taps: the per-K-step address generation is branch-free
// straight-line code that the scheduler interleaves with the MFMAs.
// branch-free "address or zero page": all element offsets are 32-bit (tensors < 2^31 floats, checked on the host)
__device__ __forceinline__ const float* sel_ptr(bool ok, const float* base, int off, const float* zero) {
  const float* b = ok ? base : zero;
  const unsigned o = ok ? (unsigned)off : 0u;
  return b + o;
}

// NARROW selects the wave arrangement / matrix instruction of the block:
//   0: 2 x 2 waves, v_mfma_f32_32x32x2_f32, block tile (64 MI) x (64 NI)
//   1: 4 x 1 waves, v_mfma_f32_16x16x4_f32, block tile (64 MI) x (48 NI): N = 48 / 96 (HRNet branch widths) without padding
//   2: 1 x 4 waves, v_mfma_f32_16x16x4_f32, block tile (48 MI) x (64 NI): the same for the M side (backward-weight, M = Cout)
//   3 / 4: as 1 with a 16 / 32 wide tile (NI = 1): per-group GEMMs of grouped convolutions (ResNeXt: 8 ... 32 channels per group)
//          and the K-class logit layers
// The LDS images and the staging code are identical in all three (64 MI rows of A, 64 NI rows / columns of B per K-step);
// the narrow forms simply do not touch the last quarter of the narrow operand.
constexpr int igemm_min_waves(int MI, int NI, int NARROW) {
  return NARROW ? ((NARROW == 1 ? MI * 3 * NI : (NARROW == 2 ? 3 * MI * NI : MI * (NARROW - 2))) <= 12 ? 4 : 2)
                : ((MI * NI == 8) ? 2 : ((MI == 2 && NI == 2) ? 4 : 1));
}

template <int LAYOUT, int MI, int NI, bool FAST, int NARROW>
__device__ __forceinline__ void igemm_f32_body(const IgemmArgs& p) {
  constexpr int BM = 64 * MI, BN = 64 * NI;                    // rows of A / B staged in LDS per K-step
  constexpr int TILE_M = NARROW == 2 ? 48 * MI : BM;           // extent of the output tile
  constexpr int TILE_N = NARROW == 1 ? 48 * NI : (NARROW == 3 ? 16 : (NARROW == 4 ? 32 : BN));
  constexpr int TM16 = NARROW == 2 ? 3 * MI : MI;              // 16 x 16 accumulator tiles per wave (narrow forms)
  constexpr int TN16 = NARROW == 2 ? NI : TILE_N / 16;
  constexpr bool A_KC = (LAYOUT != L_TN);
  constexpr bool B_KC = (LAYOUT == L_NT);
  // K sub-steps (of 16) per barrier interval.  Measured: > 1 on the small tiles costs more in occupancy
  // (LDS per block) than it saves in barriers (48-channel 3x3: 67 -> 56 TFLOP/s with 4), so it stays 1.
  constexpr int KSUB = 1;
  // LDS ring: 3 slots = ONE barrier per K-step (the slot refilled after the barrier of step k was last read in
  // step k-1, which every wave has left); 2 slots = two barriers.  Small tiles keep 2 (occupancy).
  constexpr int NBUF = (CATSEG_NBUF_BIG > 2 && (MI * NI >= 4)) || (CATSEG_NBUF_NARROW > 2 && NARROW != 0 && NARROW != 2) ? 3 : 2;
  constexpr int SLAB = (BM + BN) * 16;  // floats of one (A, B) sub-step image
  __shared__ __attribute__((aligned(16))) float smem[NBUF * KSUB * SLAB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
  // contiguous run of tiles (bijective for any grid size).
  const int nblk = p.tilesM * p.tilesN, bid = blockIdx.x;   // (= gridDim.x, except in the multi-problem launch)
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = swz / p.tilesN, tile_n = swz - tile_m * p.tilesN;
  const int m0 = tile_m * TILE_M, n0 = tile_n * TILE_N;

  int zb = blockIdx.z, split = 0;
  if (LAYOUT == L_TN) {
    split = zb % p.splits;
    zb = zb / p.splits;
  }
  const float* gbase = p.g.base + zb * p.g_bs;
  const float* obase = p.other + zb * p.o_bs;
  float* cbase = p.C + zb * p.c_bs;

  f32x16 acc[NARROW ? 1 : MI][NARROW ? 1 : NI];
  f32x4 acc16[NARROW ? TM16 : 1][NARROW ? TN16 : 1];
  if (NARROW == 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[NARROW ? 0 : i][NARROW ? 0 : j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < TM16; ++i)
#pragma unroll
      for (int j = 0; j < TN16; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[NARROW ? i : 0][NARROW ? j : 0][r] = 0.f;
  }
  // TWO-LEVEL ACCUMULATION (round 4; forward convolutions on the tiles with <= 32 accumulator registers): every second K-step (32
  // reduction elements) the running accumulator is added to a second one and cleared.  A plain fp32 FMA chain over K elements of one
  // sign (post-ReLU activations) has a rounding error ~ K / sqrt(2) ulps of a term; chains of 32 inside a chain of K / 32 cut that 2 - 3x
  // (emulated on the CPU for the stem's 64 -> 64 3x3 / 2, K = 576, against fp64: plain chain 3.3e-7 of the output RMS, oneDNN's blocked
  // sums 1.65e-7, this scheme 1.1e-7).  It matters because the FIRST layers' error is what the 300-layer network amplifies 250x into the
  // logits (tools/error_growth.py: stem 4.3e-7 against the CPU's 3.0e-7, and the ratio 1.4 - 1.5 stays to the end).
  // Forward only: in backward-data / backward-weight the second accumulator set costs occupancy (fp32 backward-weight population of the
  // HRNet-W48 step 59 -> 45 TFLOP/s when it was tried there) and the logit error does not depend on them.
  constexpr bool TWO_LEVEL = LAYOUT == L_NT && (NARROW ? TM16 * TN16 * 4 <= 32 : MI * NI * 16 <= 32);
  f32x16 tot[TWO_LEVEL && !NARROW ? MI : 1][TWO_LEVEL && !NARROW ? NI : 1];
  f32x4 tot16[TWO_LEVEL && NARROW ? TM16 : 1][TWO_LEVEL && NARROW ? TN16 : 1];
  if (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < (NARROW ? 1 : MI); ++i)
#pragma unroll
      for (int j = 0; j < (NARROW ? 1 : NI); ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[TWO_LEVEL && !NARROW ? i : 0][TWO_LEVEL && !NARROW ? j : 0][r] = 0.f;
#pragma unroll
    for (int i = 0; i < (NARROW ? TM16 : 1); ++i)
#pragma unroll
      for (int j = 0; j < (NARROW ? TN16 : 1); ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot16[TWO_LEVEL && NARROW ? i : 0][TWO_LEVEL && NARROW ? j : 0][r] = 0.f;
  }
  auto flush = [&](const bool clear) {     // tot += acc (; acc = 0)
    if (!TWO_LEVEL) return;
    if (NARROW == 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            f32x16& a = acc[NARROW ? 0 : i][NARROW ? 0 : j];
            f32x16& t = tot[TWO_LEVEL && !NARROW ? i : 0][TWO_LEVEL && !NARROW ? j : 0];
            if (clear) {
              t += a;
#pragma unroll
              for (int r = 0; r < 16; ++r) a[r] = 0.f;
            } else a += t;
          }
    } else {
#pragma unroll
      for (int i = 0; i < TM16; ++i)
#pragma unroll
        for (int j = 0; j < TN16; ++j) {
            f32x4& a = acc16[NARROW ? i : 0][NARROW ? j : 0];
            f32x4& t = tot16[TWO_LEVEL && NARROW ? i : 0][TWO_LEVEL && NARROW ? j : 0];
            if (clear) {
              t += a;
              a = f32x4{0.f, 0.f, 0.f, 0.f};
            } else a += t;
          }
    }
  };
  // narrow forms: lane = (i16, g): row / column i16 of a 16 x 16 tile, k-group g; wave origin inside the block tile
  const int i16 = lane & 15, g16 = lane >> 4;
  // k-chunk read by lane group g: {0, 3, 1, 2}.  ds_read_b128 is banked over four NON-contiguous 16-lane groups
  // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...): with the identity assignment the rows 0-3 / 4-7 (and 8-11 / 12-15) of a group
  // land on the same 16-byte slot of the (row>>2)&3-swizzled image (2-way conflict, SQ_LDS_BANK_CONFLICT = 50 % of the LDS
  // cycles); this permutation gives every group four distinct slots.  A and B use the same k assignment, so the MFMA still
  // contracts matching k.
  const int kq16 = (0x9C >> (2 * g16)) & 3;
  const int wrow16 = NARROW != 2 ? wave * 16 * MI : 0;
  const int wcol16 = NARROW == 2 ? wave * 16 * NI : 0;

  // ---- per-thread staging state --------------------------------------------------------
  // K-contiguous tiles: thread handles chunk position (tid & 3) of row j*64 + (tid >> 2).
  const int kc_row = tid >> 2;
  const int kc_chunk = (tid & 3) ^ ((tid >> 4) & 3);  // logical chunk stored at position tid & 3
  // Fast gather: per row, the pixel offset of tap (0,0) and a bit mask of the taps that fall inside
  // the image are computed once; per K-step the address is base + row offset + (uniform) tap offset.
  constexpr bool fast = FAST;
  int ab[MI], ay[MI], ax[MI];
  bool arv[MI];
  int aoff[MI];
  unsigned amask[MI];
  if (A_KC) {
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int r = m0 + j * 64 + kc_row;
      arv[j] = r < p.g.rows;
      ab[j] = ay[j] = ax[j] = 0;
      aoff[j] = 0;
      amask[j] = 0;
      if (fast || p.g.mode != 0) decode_row(p.g, arv[j] ? r : 0, ab[j], ay[j], ax[j]);
      if (fast && arv[j]) {
        const int y0 = ay[j] * p.row_s + p.row_o, x0 = ax[j] * p.row_s + p.row_o;
        aoff[j] = ((ab[j] * p.g.H + y0) * p.g.W + x0) * p.g.ld;
        int t = 0;
        for (int i = 0; i < p.nky; ++i)
          for (int jx = 0; jx < p.nkx; ++jx, ++t) {
            const int yy = y0 + p.dy0 + i * p.ddy, xx = x0 + p.dx0 + jx * p.ddx;
            if ((unsigned)yy < (unsigned)p.g.H && (unsigned)xx < (unsigned)p.g.W) amask[j] |= 1u << t;
          }
      }
    }
  }

  int nks, r_begin = 0, r_end = 0;
  const int nck = (p.Cred + 15) >> 4;
  if (LAYOUT == L_TN) {
    r_begin = split * p.rows_per_split;
    r_end = min(r_begin + p.rows_per_split, p.g.rows);
    nks = (max(r_end - r_begin, 0) + 15) >> 4;
  } else {
    nks = p.taps * nck;
  }
  int tky = 0, tkx = 0, ttap = 0, tck = 0, tks = 0;  // staging cursor (wave-uniform)
  if (LAYOUT == L_TN) {
    ttap = blockIdx.y;
    tky = (p.g.mode == 3) ? ttap : ttap / p.g.kw;
    tkx = (p.g.mode == 3) ? 0 : ttap - tky * p.g.kw;
  }

  const float* pa[KSUB][MI];
  const float* pb[KSUB][NI];

  // L_TN FAST: per staged B row the pixel (b, y, x) is decoded once and then advanced by 16 rows per K-step;
  // the filter tap / channel of the lane's column block is loop invariant
  int tb[NI], ty[NI], tx[NI], tr[NI], tky_l[NI], tkx_l[NI], tch[NI];
  bool tcv[NI];
  if (LAYOUT == L_TN && FAST) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int q = j * 256 + tid;
      const int krow = q / (16 * NI), cc = q % (16 * NI);
      tr[j] = r_begin + krow;
      decode_row(p.g, tr[j] < p.g.rows ? tr[j] : 0, tb[j], ty[j], tx[j]);
      const int col = ((n0 >> 2) + cc) * 4;
      const int tap = col / p.tap_cin;
      tch[j] = col - tap * p.tap_cin;
      tky_l[j] = tap / p.g.kw;
      tkx_l[j] = tap - tky_l[j] * p.g.kw;
      tcv[j] = col < ((p.N + 3) & ~3);
    }
  }

  // source addresses of a future K sub-step (pure VALU/SALU work: overlaps the MFMAs of the current one)
  auto prep = [&](const int sub) {
    const bool live = tks < nks;  // sub-steps past the end of the reduction load zeros
    if (LAYOUT != L_TN) {
      ++tks;
      const int chunk = tck * 4 + kc_chunk;
      const bool cv = live && chunk * 4 < p.Cred;
      const int wt = FAST ? (p.wy0 + tky * p.wys) * p.g.kw + p.wx0 + tkx * p.wxs : ttap;
      if (fast) {
        const int toff = ((p.dy0 + tky * p.ddy) * p.g.W + (p.dx0 + tkx * p.ddx)) * p.g.ld + chunk * 4;
#pragma unroll
        for (int j = 0; j < MI; ++j)
#ifdef IGEMM_BLOCKED_AB   // address-only timing experiment (tools/ab_igemm.sh): activations laid out [C/16][pixel][16] fp32; values are garbage
          pa[sub][j] = sel_ptr(cv && ((amask[j] >> (ttap & 31)) & 1u), gbase,
                               (tck * p.M + aoff[j] / p.g.ld + (p.dy0 + tky * p.ddy) * p.g.W + (p.dx0 + tkx * p.ddx)) * 16 + kc_chunk * 4, p.zero);
#else
          pa[sub][j] = sel_ptr(cv && ((amask[j] >> (ttap & 31)) & 1u), gbase, aoff[j] + toff, p.zero);
#endif
      } else {
#pragma unroll
        for (int j = 0; j < MI; ++j)
          pa[sub][j] = gather_ptr(p.g, gbase, arv[j], m0 + j * 64 + kc_row, ab[j], ay[j], ax[j], tky, tkx, chunk, cv);
      }
      if (B_KC) {
        const int boff = wt * p.tap_stride + chunk * 4;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = n0 + j * 64 + kc_row;
#ifdef IGEMM_BLOCKED_AB   // weights laid out [K/16][N][16]
          pb[sub][j] = sel_ptr(n < p.N && cv, obase, ((wt * nck + tck) * p.N + n) * 16 + kc_chunk * 4, p.zero);
#else
          pb[sub][j] = sel_ptr(n < p.N && cv, obase, n * p.ldo + boff, p.zero);
#endif
        }
      } else {
        const int boff = wt * p.tap_stride + n0;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int q = j * 256 + tid;
          const int krow = q / (16 * NI), cpos = q % (16 * NI);
          // 16x16x4 forms: lanes 0-15 / 16-31 of a ds_read_b32 read rows k and k + 4 of this K-major image, 64 NI floats apart = the
          // same banks.  Rows with (k >> 2) & 1 are therefore stored rotated by 16 floats (the LDS-DMA destination is lane-linear:
          // the lane at position cpos fetches the column chunk that belongs there): conflict-free reads.
          const int cc = (LAYOUT == L_NN && NARROW == 1) ? ((cpos - 4 * ((krow >> 2) & 1)) & (16 * NI - 1)) : cpos;
          const int k = tck * 16 + krow;
          const bool ok = live && k < p.Cred_b && (n0 + cc * 4) < ((p.N + 3) & ~3);
          pb[sub][j] = sel_ptr(ok, obase, k * p.ldo + boff + cc * 4, p.zero);
        }
      }
      if (FAST) {  // branch-free cursor advance
        const int nt = tck + 1, nx = tkx + 1;
        const bool wrap = nt == nck, wrapx = wrap && (nx == p.nkx);
        tck = wrap ? 0 : nt;
        ttap += wrap ? 1 : 0;
        tkx = wrap ? (wrapx ? 0 : nx) : tkx;
        tky += wrapx ? 1 : 0;
      } else if (++tck == nck) {
        tck = 0;
        ++ttap;
        if (p.g.mode == 3) {
          ++tky;
        } else if (++tkx == p.g.kw) {
          tkx = 0;
          ++tky;
        }
      }
    } else {
      const int kbase = r_begin + tks * 16;
      ++tks;
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int q = j * 256 + tid;
        const int krow = q / (16 * MI), cc = q % (16 * MI);
        const int r = kbase + krow;
        const bool ok = r < r_end && (m0 + cc * 4) < ((p.M + 3) & ~3);
        pa[sub][j] = sel_ptr(ok, obase, r * p.ldo + m0 + cc * 4, p.zero);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int q = j * 256 + tid;
        const int krow = q / (16 * NI), cc = q % (16 * NI);
        const int r = kbase + krow;
        const bool rv = r < r_end;
        int b = 0, y = 0, x = 0;
        const int chunk = (n0 >> 2) + cc;
        const bool cvv = chunk * 4 < ((p.N + 3) & ~3);
        if (FAST) {  // forward-conv gather (plain GEMM rows are passed as a degenerate 1-wide image)
          const int iy = ty[j] * p.g.stride - p.g.pad + tky_l[j] * p.g.dil;
          const int ix = tx[j] * p.g.stride - p.g.pad + tkx_l[j] * p.g.dil;
          const bool ok = tr[j] < r_end && tcv[j] && (unsigned)iy < (unsigned)p.g.H && (unsigned)ix < (unsigned)p.g.W;
          pb[sub][j] = sel_ptr(ok, gbase, ((tb[j] * p.g.H + iy) * p.g.W + ix) * p.g.ld + tch[j], p.zero);
          // advance this row by 16 pixels
          tr[j] += 16;
          tx[j] += p.step_rx;
          ty[j] += p.step_qy;
          const bool cx = tx[j] >= p.g.Wo;
          tx[j] -= cx ? p.g.Wo : 0;
          ty[j] += cx ? 1 : 0;
          const bool cy = ty[j] >= p.g.Ho;
          ty[j] -= cy ? p.g.Ho : 0;
          tb[j] += p.step_b + (cy ? 1 : 0);
        } else {
          if (p.g.mode != 0) decode_row(p.g, rv ? r : 0, b, y, x);
          pb[sub][j] = gather_ptr(p.g, gbase, rv, r, b, y, x, tky, tkx, chunk, cvv);
        }
      }
    }
  };

  auto issue = [&](const int buf, const int sub) {
    float* sA = smem + (buf * KSUB + sub) * SLAB;
    float* sB = sA + BM * 16;
#pragma unroll
    for (int j = 0; j < MI; ++j) glds16(pa[sub][j], sA + (j * 256 + wave * 64) * 4);
#pragma unroll
    for (int j = 0; j < NI; ++j) glds16(pb[sub][j], sB + (j * 256 + wave * 64) * 4);
  };

  auto compute = [&](const int buf, const int sub, const bool more) {
    const float* sA = smem + (buf * KSUB + sub) * SLAB;
    const float* sB = sA + BM * 16;
    if constexpr (NARROW != 0) {
      // one read round per K-step: lane (i16, g) fetches k = 4g .. 4g+3 of its row / column; MFMA e consumes k = 4g + e
      float a[TM16][4], b[TN16][4];
#pragma unroll
      for (int t = 0; t < TM16; ++t) {
        const int row = wrow16 + t * 16 + i16;
        if (A_KC) {
          const int pos = kq16 ^ ((row >> 2) & 3);
          const f32x4 v = *(const f32x4*)(sA + row * 16 + pos * 4);
          a[t][0] = v[0]; a[t][1] = v[1]; a[t][2] = v[2]; a[t][3] = v[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) a[t][e] = sA[(4 * kq16 + e) * BM + row];
        }
      }
#pragma unroll
      for (int u = 0; u < TN16; ++u) {
        const int col = wcol16 + u * 16 + i16;
        if (B_KC) {
          const int pos = kq16 ^ ((col >> 2) & 3);
          const f32x4 v = *(const f32x4*)(sB + col * 16 + pos * 4);
          b[u][0] = v[0]; b[u][1] = v[1]; b[u][2] = v[2]; b[u][3] = v[3];
        } else {
          const int colr = (LAYOUT == L_NN && NARROW == 1) ? ((col + 16 * (kq16 & 1)) & (BN - 1)) : col;   // rows 4 kq16 + e: (k >> 2) & 1 = kq16 & 1
#pragma unroll
          for (int e = 0; e < 4; ++e) b[u][e] = sB[(4 * kq16 + e) * BN + colr];
        }
      }
      if (FAST || more) prep(sub);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < TM16; ++t)
#pragma unroll
          for (int u = 0; u < TN16; ++u)
            acc16[NARROW ? t : 0][NARROW ? u : 0] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][e], b[u][e], acc16[NARROW ? t : 0][NARROW ? u : 0], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      float a[MI][4], b[NI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * 32 * MI + mi * 32 + l31;
        if (A_KC) {
          const int pos = (2 * jj + h) ^ ((row >> 2) & 3);
          const f32x4 v = *(const f32x4*)(sA + row * 16 + pos * 4);
          a[mi][0] = v[0]; a[mi][1] = v[1]; a[mi][2] = v[2]; a[mi][3] = v[3];
        } else {
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) a[mi][ii] = sA[(8 * jj + 4 * h + ii) * BM + row];
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * 32 * NI + ni * 32 + l31;
        if (B_KC) {
          const int pos = (2 * jj + h) ^ ((row >> 2) & 3);
          const f32x4 v = *(const f32x4*)(sB + row * 16 + pos * 4);
          b[ni][0] = v[0]; b[ni][1] = v[1]; b[ni][2] = v[2]; b[ni][3] = v[3];
        } else {
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) b[ni][ii] = sB[(8 * jj + 4 * h + ii) * BN + row];
        }
      }
      __builtin_amdgcn_iglp_opt(0);  // scheduler hint: interleave the LDS reads with the MFMA chain (+1 % measured)
      if (jj == 0 && (FAST || more)) prep(sub);  // address math for the interval after next rides under the MFMAs
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[NARROW ? 0 : mi][NARROW ? 0 : ni] =
                __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][ii], b[ni][ii], acc[NARROW ? 0 : mi][NARROW ? 0 : ni], 0, 0, 0);
    }
  };

  // ---- main loop: K-step k+1 is in flight (LDS-DMA) while k is computed; addresses of k+2 are
  // prepared inside the MFMA region ------------------------------------------------------------
  if (nks > 0 && NBUF == 2) {
    const int nss = (nks + KSUB - 1) / KSUB;  // barrier intervals
#pragma unroll
    for (int sub = 0; sub < KSUB; ++sub) {
      prep(sub);
      issue(0, sub);
    }
#pragma unroll
    for (int sub = 0; sub < KSUB; ++sub) prep(sub);
    for (int ss = 0; ss < nss; ++ss) {
      const int cur = ss & 1;
      if (ss + 1 < nss) {
#pragma unroll
        for (int sub = 0; sub < KSUB; ++sub) issue(cur ^ 1, sub);
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(KSUB * (MI + NI)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int sub = 0; sub < KSUB; ++sub) compute(cur, sub, ss + 2 < nss);
      if (TWO_LEVEL && (ss & 1)) flush(true);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else if (nks > 0) {
    // three-slot ring, one barrier per K-step: steps k+1 and k+2 are in flight while k is computed
    prep(0);
    issue(0, 0);
    prep(0);
    if (nks > 1) issue(1, 0);
    prep(0);  // addresses of step 2
    int cur = 0, fill = 2;
    for (int ks = 0; ks < nks; ++ks) {
      if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(MI + NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (ks + 2 < nks) issue(fill, 0);
      compute(cur, 0, ks + 3 < nks);
      if (TWO_LEVEL && (ks & 1)) flush(true);
      cur = cur == 2 ? 0 : cur + 1;
      fill = fill == 2 ? 0 : fill + 1;
    }
  }
  flush(false);     // acc = tot + (the last, possibly odd, K-step)

  // ---- epilogue -------------------------------------------------------------------------
  float* cout = cbase;
  if (LAYOUT == L_TN) cout += split * p.c_split_stride + (long long)blockIdx.y * p.c_tap_stride;
  // ---- FAST PATH (round 6): a tile wholly inside M x N, no residual operand and no row remap -- every tile of most layers this kernel
  // still takes (M a multiple of the tile height, N of its width).  The general path below predicates every one of the tile's stores with
  // scalar branches, computes a 64-bit address per element and selects the valid rows of the BatchNorm partials: ~2500 instructions per block
  // for the (128 x 64) tile of <NT, 2, 1> against ~500 in its whole K loop when K = 64 (the stage-1 pointwise layers): the kernels were
  // instruction-bound in their epilogue (csrc/pconv1.hip, same finding, SQ counters in profiles/r06_pmc_sq_p1_*.json).  Here: the same
  // values ((acc + bias), ReLU) through unconditional buffer stores -- one per-lane offset, the accumulator row in the scalar offset, the
  // column tile in the immediate -- and partial sums without selects.  Only where the accumulators leave room for a second epilogue in the
  // register budget (MI * NI <= 2; backward-data / backward-weight, which carry no second accumulator set, up to 4).
  if constexpr (NARROW == 0 && (MI * NI <= 2 || (LAYOUT != L_NT && MI * NI <= 4))) {
    const bool has_res = LAYOUT == L_NT && p.residual != nullptr;
    if (m0 + TILE_M <= p.M && n0 + TILE_N <= p.N && !has_res && !(LAYOUT != L_TN && p.remap)) {
      float bvv[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bvv[ni] = p.bias != nullptr ? p.bias[zb * p.bias_bs + n0 + wn * 32 * NI + ni * 32 + l31] : 0.f;
      if (LAYOUT == L_NT && p.bn_part != nullptr) {
        __syncthreads();   // every wave has left the K loop: its LDS buffers are free
        int colv[NI];
        float bvb[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          colv[ni] = wn * 32 * NI + ni * 32 + l31;
          bvb[ni] = p.bias != nullptr ? p.bias[n0 + colv[ni]] : 0.f;
        }
        cs_tile_bn_partials<NI, MI * 16, 2, false>(
            smem, TILE_N, colv, h == 0, wm, TILE_M, [&](int j, int i) { return acc[i >> 4][j][i & 15] + bvb[j]; }, [&](int) { return true; },
            p.bn_part + (long long)tile_m * 3 * p.N, p.N, n0);
      }
      const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(cout + (long long)m0 * p.ldc + n0), (short)0,
                                                                          (int)((unsigned)TILE_M * (unsigned)p.ldc * 4u), 0x00020000);
      const int ld4 = p.ldc * 4;
      const int voff = (wm * 32 * MI + 4 * h) * ld4 + (wn * 32 * NI + l31) * 4;
      const bool relu = LAYOUT == L_NT && p.relu, accum = p.accumulate != 0;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int soff = (mi * 32 + (r & 3) + 8 * (r >> 2)) * ld4;
          float v[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) v[ni] = acc[mi][ni][r] + bvv[ni];
          if (accum) {                  // ((acc + bias) + previous contents, as the general path; a row is read before it is stored)
            float o[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) o[ni] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsC, voff + 128 * ni, soff, 0));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) v[ni] += o[ni];
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (relu) v[ni] = fmaxf(v[ni], 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[ni]), rsC, voff + 128 * ni, soff, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
  }
  auto out_row = [&](const int row) {
    long long orow = row;
    if (LAYOUT != L_TN && p.remap) {
      int b, a, c;
      decode_row(p.g, row, b, a, c);
      orow = ((long long)b * p.out_H + a * p.out_s + p.out_py) * p.out_W + c * p.out_s + p.out_px;
    }
    return orow;
  };
  // what is ADDED to an element (its previous contents when accumulating, the residual branch): the callers read it for a whole
  // accumulator tile before the first store -- a load behind a store that may alias it cannot be hoisted and waits for its own
  // round trip (1x1 256 <- 64 backward-data accumulating into the block input's gradient: 322 us against 136 us without)
  auto fetch = [&](const int row, const long long orow, const int col) {
    float a = 0.f;
    if (row < p.M && col < p.N) {
      if (p.accumulate) a = cout[orow * p.ldc + col];
      if (LAYOUT == L_NT && p.residual) a += p.residual[orow * p.ldr + col];
    }
    return a;
  };
  const bool adds = p.accumulate || (LAYOUT == L_NT && p.residual != nullptr);
  auto store = [&](const int row, const long long orow, const int col, const float av, const float bv, const float add) {
    if (row < p.M) {
      float* dst = cout + orow * p.ldc + col;
      if (col < p.N) {
        float v = (av + bv) + add;
        if (LAYOUT == L_NT && p.relu) v = fmaxf(v, 0.f);
        *dst = v;
      } else if (col < p.zero_to) {
        *dst = 0.f;
      }
    }
  };
  // BatchNorm batch statistics of this tile (training forward): see cs_tile_bn_partials in common.h
  if (LAYOUT == L_NT && (NARROW == 0 || NARROW == 1) && p.bn_part != nullptr) {
    __syncthreads();   // every wave has left the K loop: its LDS buffers are free
    const int nvalid = min(TILE_M, p.M - m0);
    float* part = p.bn_part + (long long)tile_m * 3 * p.N;
    if constexpr (NARROW == 0) {
      int colv[NI];
      float bv[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        colv[ni] = wn * 32 * NI + ni * 32 + l31;
        bv[ni] = (p.bias != nullptr && n0 + colv[ni] < p.N) ? p.bias[n0 + colv[ni]] : 0.f;
      }
      const int rbase = m0 + wm * 32 * MI + 4 * h;
      cs_tile_bn_partials<NI, MI * 16, 2, false>(
          smem, TILE_N, colv, h == 0, wm, nvalid,
          [&](int j, int i) { return acc[NARROW ? 0 : (i >> 4)][NARROW ? 0 : j][i & 15] + bv[j]; },
          [&](int i) { return rbase + (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2) < p.M; }, part, p.N, n0);
    } else {
      int colv[TN16];
      float bv[TN16];
#pragma unroll
      for (int u = 0; u < TN16; ++u) {
        colv[u] = u * 16 + i16;
        bv[u] = (p.bias != nullptr && n0 + colv[u] < p.N) ? p.bias[n0 + colv[u]] : 0.f;
      }
      const int rbase = m0 + wrow16 + 4 * g16;
      cs_tile_bn_partials<TN16, TM16 * 4, 4, true>(
          smem, TILE_N, colv, g16 == 0, wave, nvalid,
          [&](int j, int i) { return acc16[NARROW ? (i >> 2) : 0][NARROW ? j : 0][i & 3] + bv[j]; },
          [&](int i) { return rbase + (i >> 2) * 16 + (i & 3) < p.M; }, part, p.N, n0);
    }
  }
  if constexpr (NARROW != 0) {
#pragma unroll
    for (int t = 0; t < TM16; ++t) {
#pragma unroll
      for (int u = 0; u < TN16; ++u) {
        const int col = n0 + wcol16 + u * 16 + i16;
        const float bv = (p.bias != nullptr && col < p.N) ? p.bias[zb * p.bias_bs + col] : 0.f;
        float add[4] = {0.f, 0.f, 0.f, 0.f};
        int orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) orow[r] = (int)out_row(m0 + wrow16 + t * 16 + 4 * g16 + r);
        if (adds)
#pragma unroll
          for (int r = 0; r < 4; ++r) add[r] = fetch(m0 + wrow16 + t * 16 + 4 * g16 + r, orow[r], col);
#pragma unroll
        for (int r = 0; r < 4; ++r) store(m0 + wrow16 + t * 16 + 4 * g16 + r, orow[r], col, acc16[NARROW ? t : 0][NARROW ? u : 0][r], bv, add[r]);
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      int orow16[16];              // output row of every accumulator row (strided backward-data remaps them: one decode per row, not per column tile)
#pragma unroll
      for (int r = 0; r < 16; ++r) orow16[r] = (int)out_row(m0 + wm * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = n0 + wn * 32 * NI + ni * 32 + l31;
        const float bv = (p.bias != nullptr && col < p.N) ? p.bias[zb * p.bias_bs + col] : 0.f;
        float add[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = 0.f;
        if (adds)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            add[r] = fetch(row, orow16[r], col);
          }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          store(m0 + wm * 32 * MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, orow16[r], col, acc[NARROW ? 0 : mi][NARROW ? 0 : ni][r], bv, add[r]);
      }
    }
  }
}

template <int LAYOUT, int MI, int NI, bool FAST, int NARROW>
__global__ __launch_bounds__(256, igemm_min_waves(MI, NI, NARROW)) void igemm_f32_kernel(const IgemmArgs p) {
  igemm_f32_body<LAYOUT, MI, NI, FAST, NARROW>(p);
}

// Strided backward-data: the (up to four) input-pixel parity classes of one convolution as ONE launch, blockIdx.y = class.  The
// classes are independent dense problems with their own tap sets (1 / 2 / 2 / 4 taps of a 3x3 stride-2 filter) and row counts;
// launched one after the other (round 1) each of them filled a fraction of the chip and paid its own launch.
struct IgemmMulti { IgemmArgs a[4]; };
template <int MI, int NI, int NARROW>
__global__ __launch_bounds__(256, igemm_min_waves(MI, NI, NARROW)) void igemm_f32_multi_kernel(const IgemmMulti q) {
  const IgemmArgs& p = q.a[blockIdx.y];
  if ((int)blockIdx.x >= p.tilesM * p.tilesN) return;
  igemm_f32_body<L_NN, MI, NI, true, NARROW>(p);
}

// out[i] = sum_s slab[s][i]  (deterministic split reduction of backward-weight partials)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n4,
                                    int splits, long long stride4) {
  const f32x4* s = (const f32x4*)slabs;
  f32x4* o = (f32x4*)out;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 a = s[i];
    for (int k = 1; k < splits; ++k) a += s[i + k * stride4];
    o[i] = a;
  }
}

// the same for MANY slabs of a small tensor (direct backward-weight: up to 512 slabs of 83-330 KB): a block owns 64
// float4 columns, its 16 waves each add a contiguous range of slabs (8 independent loads in flight), the 16 partials
// are combined through LDS in wave order -> fixed summation order, deterministic
__global__ __launch_bounds__(1024) void reduce_slabs_wide_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n4,
                                                                 int splits, long long stride4) {
  const f32x4* s = (const f32x4*)slabs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + lane;
  const int per = (splits + 15) / 16;
  const int k0 = wave * per, k1 = min(k0 + per, splits);
  f32x4 a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += s[i + (long long)(k + j) * stride4];
    }
    for (; k < k1; ++k) a[0] += s[i + (long long)k * stride4];
  }
  __shared__ f32x4 part[16][64];
  part[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (wave == 0 && i < n4) {
    f32x4 t = part[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += part[w][lane];
    ((f32x4*)out)[i] = t;
  }
}

// dbias[o] = sum_p dy[p, o]: block = 64 channels x 4 row lanes, rows strided over gridDim.y
__global__ void colsum_partial_kernel(const float* __restrict__ dy, int ld, long long rows, int C,
                                      float* __restrict__ part) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (long long r = blockIdx.y * 4 + rl; r < rows; r += (long long)gridDim.y * 4) s += dy[r * ld + c];
  __shared__ float sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < C) part[(long long)blockIdx.y * C + c] = sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192];
}
// the K-class logits' gradient (rows of 32 floats, C <= 32 valid columns; models/OCR.py:82-85, 98): 16-byte loads, 8 column quads x 32 row lanes per
// block (the generic kernel above keeps 25 of 64 lanes busy with 4-byte loads 128 bytes apart: 90 us for a 33 MB tensor)
__global__ __launch_bounds__(256) void colsum_ld32_kernel(const float* __restrict__ dy, long long rows, float* __restrict__ part) {
  const int q = threadIdx.x & 7, rl = threadIdx.x >> 3;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  long long r = (long long)blockIdx.x * 32 + rl;
  const long long st = (long long)gridDim.x * 32;
  for (; r + st < rows; r += 2 * st) {
    s0 += *(const f32x4*)(dy + r * 32 + q * 4);
    s1 += *(const f32x4*)(dy + (r + st) * 32 + q * 4);
  }
  if (r < rows) s0 += *(const f32x4*)(dy + r * 32 + q * 4);
  __shared__ f32x4 sh[256];
  sh[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (rl == 0) {
    f32x4 t = sh[q];
    for (int k = 1; k < 32; ++k) t += sh[k * 8 + q];
    *(f32x4*)(part + (long long)blockIdx.x * 32 + q * 4) = t;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out, int n_out = -1) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (n_out < 0 ? C : n_out)) return;
  // four independent chains (up to 256 partial rows: one dependent chain of loads was 66 us on the step's serial path), combined in a fixed order
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = 0;
  for (; i + 3 < nparts; i += 4) {
    s0 += part[(long long)i * C + c];
    s1 += part[(long long)(i + 1) * C + c];
    s2 += part[(long long)(i + 2) * C + c];
    s3 += part[(long long)(i + 3) * C + c];
  }
  for (; i < nparts; ++i) s0 += part[(long long)i * C + c];
  out[c] = (s0 + s1) + (s2 + s3);
}

int g_force_mi = 0, g_force_ni = 0, g_force_narrow = 0, g_force_splits = 0, g_strided_multi = 1;

}  // namespace
// wgrad_direct.hip
size_t wgrad_direct_workspace(const catseg_conv_desc* d);
int wgrad_direct_launch(const catseg_conv_desc* d, const float* x, const float* dy, void* workspace, hipStream_t st);
namespace {
  // tuning hooks (catseg_debug_set_tile / _splits)

// address of the zero page in the CURRENT device's copy of the code object (a __device__ symbol has one instance per device)
const float* zero_page_ptr() {
  static const float* z[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!z[dev]) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_page)) == hipSuccess) z[dev] = (const float*)q;
  }
  return z[dev];
}

template <int LAYOUT, int MI, int NI, int NARROW = 0>
void launch_one(const IgemmArgs& a0, int ncols, int nbatch, int grid_y, hipStream_t st) {
  IgemmArgs a = a0;
  a.zero = zero_page_ptr();
  constexpr int TILE_M = NARROW == 2 ? 48 * MI : 64 * MI;
  constexpr int TILE_N = NARROW == 1 ? 48 * NI : (NARROW == 3 ? 16 : (NARROW == 4 ? 32 : 64 * NI));
  a.tilesM = (a.M + TILE_M - 1) / TILE_M;
  a.tilesN = (ncols + TILE_N - 1) / TILE_N;
  dim3 grid(a.tilesM * a.tilesN, grid_y, nbatch * (LAYOUT == L_TN ? a.splits : 1));
  // 16 rows = step_b images + step_qy image rows + step_rx pixels (step_qy < Ho, step_rx < Wo: one carry each per step)
  const int img = a.g.Ho * a.g.Wo > 0 ? a.g.Ho * a.g.Wo : 1, wo = a.g.Wo > 0 ? a.g.Wo : 1;
  a.step_b = 16 / img;
  a.step_qy = (16 % img) / wo;
  a.step_rx = (16 % img) % wo;
  // TN fast path: every forward-conv gather (mode 1); rows are advanced incrementally in the K loop
  const bool fast = LAYOUT == L_TN ? (a.g.mode == 1) : (a.taps <= 32 && a.row_s > 0);
  if (fast) hipLaunchKernelGGL((igemm_f32_kernel<LAYOUT, MI, NI, true, NARROW>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((igemm_f32_kernel<LAYOUT, MI, NI, false, NARROW>), grid, dim3(256), 0, st, a);
}

// measured (tools/bench_narrow.py): narrow-N forms 94-104 TFLOP/s on N = 48 / 96 (+27-36 % over the padded 64-wide tiles);
// narrow-M backward-weight forms within +-10 % of the 64-row tiles: kept for tuning runs, not chosen by the planner (eff 0)
constexpr float kEffN2 = 0.0f;

// Tile / split planner: minimise a time model
//   t = padded FLOPs / (R x tile efficiency x CU-level quantisation) [+ slab reduction for split backward-weight].
// Quantisation is counted in tiles per CU, not per resident-block slot: blocks that share a CU share its MFMA pipes, so
// a grid that gives every CU the same number of tiles is "full" whatever the residency (measured, tools/bench_tiles_small.py:
// the machine rate of a tile is within 10 % for 1 ... 6 resident blocks per CU).
struct TilePlan { int mi, ni, splits, rps, narrow; };
struct TileInfo { int mi, ni, occ; float eff; int narrow; };
// eff: measured relative MFMA efficiency of each tile (tools/bench_tiles*.py); 256x64 / 64x256 tiles exist in the
// dispatcher but never won a measurement and are not candidates.
const TileInfo kTiles[] = {{1, 1, 8, 0.80f, 0}, {2, 1, 6, 0.90f, 0}, {1, 2, 6, 0.88f, 0}, {2, 2, 4, 1.00f, 0}, {4, 2, 2, 0.98f, 0}, {2, 4, 2, 0.96f, 0},
                           // 16x16x4 forms: (64 mi) x (48 ni) for forward / backward-data, (48 mi) x (64 ni) for backward-weight
                           {2, 1, 6, 0.85f, 1}, {4, 1, 4, 0.84f, 1}, {2, 2, 4, 0.90f, 1}, {4, 2, 2, 0.80f, 1},
                           {2, 1, 6, 0.60f, 3}, {4, 1, 6, 0.62f, 3}, {2, 1, 6, 0.70f, 4}, {4, 1, 6, 0.72f, 4},
                           {1, 2, 6, kEffN2, 2}, {1, 4, 4, kEffN2, 2}, {2, 2, 4, kEffN2, 2}, {2, 4, 2, kEffN2, 2}};

inline double cu_quant(double tiles) {
  const double c = tiles / 256.0;
  if (c <= 1.0) return c;
  return c / (double)(long long)(c + 0.999999);
}

TilePlan plan_tiles(int layout, long long M, long long ncols, long long extra, long long red_rows, bool allow_narrow = true, long long kdepth = 0) {
  TilePlan best = {2, 2, 1, 0};
  double best_t = 1e300;
  const double R = 115e12;
  for (const TileInfo& t : kTiles) {
    if (g_force_mi > 0 && (t.mi != g_force_mi || t.ni != g_force_ni || t.narrow != g_force_narrow)) continue;
    if (g_force_mi == 0 && t.eff <= 0.f) continue;
    if (t.narrow && !allow_narrow) continue;
    if ((t.narrow != 2 && t.narrow != 0 && layout == L_TN) || (t.narrow == 2 && layout != L_TN)) continue;
    if (t.narrow >= 3 && layout != L_NT) continue;   // 16 / 32 wide tiles: forward only (grouped conv, class logits)
    const long long tile_m = t.narrow == 2 ? 48 * t.mi : 64 * t.mi;
    const long long tile_n = t.narrow == 1 ? 48 * t.ni : (t.narrow == 3 ? 16 : (t.narrow == 4 ? 32 : 64 * t.ni));
    const long long tm = (M + tile_m - 1) / tile_m, tn = (ncols + tile_n - 1) / tile_n;
    if ((t.narrow == 1 || t.narrow >= 3) && tn > 1 && g_force_mi == 0) continue;  // in-network the 48/96-wide forms only win when one tile spans N
    const double padded = 2.0 * (double)(tm * tile_m) * (double)(tn * tile_n) * (double)extra * (double)red_rows;
    if (layout != L_TN) {
      double eff = t.eff > 0.f ? t.eff : 1.f;
      // forward layers with a SHORT reduction (1 x 1, <= 128 input channels: the stage-1 pointwise layers) are bound by their epilogue and
      // their stores, not by the matrix pipe: the 128 x 128 tile has no full-tile epilogue (register budget) and the 64 x 128 tile writes the
      // longest row segments -- measured at 8 x 136 x 240, 64 -> 256 (tools/sweep_f32_tiles.py): (2, 2) 175 us, (2, 1) 159, (1, 1) 164, (1, 2) 141
      if (layout == L_NT && kdepth > 0 && kdepth <= 128 && t.narrow == 0 && g_force_mi == 0)
        eff = (t.mi == 1 && t.ni == 2) ? 1.0 : ((t.mi * t.ni >= 4) ? 0.75 * eff : eff);
      const double tt = padded / (R * eff * cu_quant((double)(tm * tn * extra)));
      if (tt < best_t) {
        best_t = tt;
        best = {t.mi, t.ni, 1, 0, t.narrow};
      }
      continue;
    }
    // backward-weight: the K-major operand path is latency-bound per block, so here residency matters: the grid is
    // quantised in resident-block slots and the split count fills them (measured: the CU-level model costs 5-20 %)
    long long maxs = red_rows / 512;  // at least 32 K-steps per split
    if (maxs > 96) maxs = 96;
    if (maxs < 1) maxs = 1;
    const double slots = 256.0 * t.occ;
    const double eff = t.eff <= 0.f ? 1.0 : ((t.mi == 4 && t.narrow == 0) ? 1.03 : t.eff);  // 256x128: measured 118 vs 113 TFLOP/s
    for (long long sp = 1; sp <= maxs; ++sp) {
      const double rounds = (double)(tm * tn * extra * sp) / slots;
      const double q = rounds / (double)(long long)(rounds + 0.999999);
      const double tt = padded / (R * eff * q * (1.0 - 0.002 * (double)(sp - 1)));
      if (tt < best_t) {
        best_t = tt;
        best = {t.mi, t.ni, (int)sp, 0, t.narrow};
      }
    }
  }
  if (g_force_mi > 0 && best_t >= 1e300) best = {g_force_mi, g_force_ni, 1, 0, g_force_narrow};
  if (g_force_splits > 0 && layout == L_TN) best.splits = g_force_splits;
  if (layout == L_TN) {
    best.rps = (int)(((red_rows + best.splits - 1) / best.splits + 15) / 16 * 16);
    best.splits = (int)((red_rows + best.rps - 1) / best.rps);
  }
  return best;
}

template <int LAYOUT>
int launch_igemm(const IgemmArgs& a, int nbatch, int grid_y, hipStream_t st, const TilePlan* given = nullptr) {
  const int ncols = a.zero_to > a.N ? a.zero_to : a.N;  // pad columns to be zero-filled are visited too
  const TilePlan pl = given ? *given : plan_tiles(LAYOUT, a.M, ncols, (long long)grid_y * nbatch, a.g.rows, true, (long long)a.taps * a.Cred);
  const int mi = pl.mi, ni = pl.ni;
  bool ok = false;
#define CS_FORM(F_, M_, N_)                                                          \
  if (!ok && pl.narrow == F_ && mi == M_ && ni == N_) {                              \
    launch_one<LAYOUT, M_, N_, F_>(a, ncols, nbatch, grid_y, st);                    \
    ok = true;                                                                       \
  }
  // (256x64, 64x256, 64x448, 128x448 and 256x256 tiles were measured and never won: not instantiated)
  CS_FORM(0, 1, 1) CS_FORM(0, 1, 2) CS_FORM(0, 2, 1) CS_FORM(0, 2, 2) CS_FORM(0, 4, 2) CS_FORM(0, 2, 4)
  if constexpr (LAYOUT != L_TN) {
    CS_FORM(1, 2, 1) CS_FORM(1, 4, 1) CS_FORM(1, 2, 2) CS_FORM(1, 4, 2)
  } else {
    CS_FORM(2, 1, 2) CS_FORM(2, 1, 4) CS_FORM(2, 2, 2) CS_FORM(2, 2, 4)
  }
  if constexpr (LAYOUT == L_NT) {
    CS_FORM(3, 2, 1) CS_FORM(3, 4, 1) CS_FORM(4, 2, 1) CS_FORM(4, 4, 1)
  }
#undef CS_FORM
  if (!ok) {
    catseg_set_error("igemm: unsupported tile %dx%d (form %d)", mi, ni, pl.narrow);
    return CATSEG_EINVAL;
  }
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// the parity classes of a strided backward-data as one launch (classes sorted by work, heaviest first)
int launch_multi_nn(IgemmArgs* cls, int ncls, hipStream_t st) {
  long long maxM = 0;
  for (int i = 0; i < ncls; ++i) maxM = cls[i].M > maxM ? cls[i].M : maxM;
  const TilePlan pl = plan_tiles(L_NN, maxM, cls[0].N, ncls, maxM);
  IgemmMulti q = {};
  const float* zero = zero_page_ptr();
  bool ok = false;
#define CS_FORM(F_, M_, N_)                                                                                   \
  if (!ok && pl.narrow == F_ && pl.mi == M_ && pl.ni == N_) {                                                 \
    constexpr int TILE_M = 64 * M_, TILE_N = F_ == 1 ? 48 * N_ : 64 * N_;                                     \
    int maxt = 0;                                                                                             \
    for (int i = 0; i < ncls; ++i) {                                                                          \
      q.a[i] = cls[i];                                                                                        \
      q.a[i].zero = zero;                                                                                     \
      q.a[i].tilesM = (cls[i].M + TILE_M - 1) / TILE_M;                                                       \
      q.a[i].tilesN = (cls[i].N + TILE_N - 1) / TILE_N;                                                       \
      const int t = q.a[i].tilesM * q.a[i].tilesN;                                                            \
      maxt = t > maxt ? t : maxt;                                                                             \
    }                                                                                                         \
    hipLaunchKernelGGL((igemm_f32_multi_kernel<M_, N_, F_>), dim3(maxt, ncls, 1), dim3(256), 0, st, q);       \
    ok = true;                                                                                                \
  }
  CS_FORM(0, 1, 1) CS_FORM(0, 1, 2) CS_FORM(0, 2, 1) CS_FORM(0, 2, 2) CS_FORM(0, 4, 2) CS_FORM(0, 2, 4)
  CS_FORM(1, 2, 1) CS_FORM(1, 4, 1) CS_FORM(1, 2, 2) CS_FORM(1, 4, 2)
#undef CS_FORM
  if (!ok) {
    catseg_set_error("igemm: unsupported tile %dx%d (form %d)", pl.mi, pl.ni, pl.narrow);
    return CATSEG_EINVAL;
  }
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

int check_desc(const catseg_conv_desc* d) {
  CS_REQUIRE(d != nullptr, "conv: null descriptor");
  CS_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0, "conv: bad dims");
  CS_REQUIRE(d->stride >= 1 && d->dil >= 1 && d->kh >= 1 && d->kw >= 1 && d->pad >= 0, "conv: bad geometry");
  CS_REQUIRE(d->ldx % 4 == 0 && d->ldy % 4 == 0 && d->ldx >= d->Cin && d->ldy >= d->Cout, "conv: ld must be >= C and a multiple of 4");
  const int ho = (d->H + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
  const int wo = (d->W + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
  CS_REQUIRE(ho == d->Ho && wo == d->Wo, "conv: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
  if (d->stem4) CS_REQUIRE(d->Cin == 4 && d->kw == 7 && d->ldx == 4, "conv: stem4 needs Cin == ldx == 4, kw == 7");
  else CS_REQUIRE(d->Cin % 4 == 0, "conv: Cin must be a multiple of 4 (got %d)", d->Cin);
  if (d->groups > 1)
    CS_REQUIRE(!d->stem4 && d->Cin % d->groups == 0 && d->Cout % d->groups == 0 && (d->Cin / d->groups) % 4 == 0,
               "conv: groups must divide Cin and Cout, Cin/groups a multiple of 4");
  CS_REQUIRE((long long)d->B * d->Ho * d->Wo * d->ldy < (1ll << 31) && (long long)d->B * d->H * d->W * d->ldx < (1ll << 31) &&
                 (long long)d->Cout * d->kh * d->kw * d->Cin < (1ll << 31),
             "conv: tensors must have fewer than 2^31 elements (32-bit element offsets)");
  return CATSEG_OK;
}

// affine tap parameters of a forward conv gather (sign = +1) or a stride-1 backward-data gather (sign = -1)
void fill_taps(IgemmArgs& a, int kh, int kw, int dil, int sign, int row_s, int row_o) {
  a.row_s = 0;
  if (kh * kw > 32) return;
  a.nky = kh; a.nkx = kw;
  a.dy0 = a.dx0 = 0; a.ddy = a.ddx = sign * dil;
  a.wy0 = a.wx0 = 0; a.wys = a.wxs = 1;
  a.row_s = row_s;
  a.row_o = row_o;
}

Geo fwd_geo(const catseg_conv_desc* d, const float* x) {
  Geo g;
  g.base = x; g.mode = d->stem4 ? 3 : 1; g.rows = d->B * d->Ho * d->Wo; g.ld = d->ldx;
  g.H = d->H; g.W = d->W; g.Ho = d->Ho; g.Wo = d->Wo; g.kw = d->kw; g.stride = d->stride; g.pad = d->pad; g.dil = d->dil;
  return g;
}

}  // namespace

extern "C" int catseg_debug_set_tile(int mi, int ni) {
  g_force_narrow = mi >> 4;  // mi + 16 * form: 1 = 16x16x4 tiles narrow in N, 2 = narrow in M
  mi &= 15;
  g_force_mi = mi;
  g_force_ni = ni;
  return CATSEG_OK;
}
extern "C" int catseg_debug_set_strided_multi(int on) {
  g_strided_multi = on;
  return CATSEG_OK;
}
extern "C" int catseg_debug_set_splits(int splits) {
  g_force_splits = splits;
  return CATSEG_OK;
}

extern "C" int catseg_debug_plan_conv(const catseg_conv_desc* d, int op, int* out);

extern "C" int catseg_conv2d_fwd(const catseg_conv_desc* d, const float* x, const float* w, const float* bias,
                                 float* y, int zero_to, catseg_stream_t stream) {
  if (int e = check_desc(d)) return e;
  CS_REQUIRE(cs_aligned16(x) && cs_aligned16(w) && cs_aligned16(y), "conv fwd: pointers must be 16-byte aligned");
  CS_REQUIRE(zero_to <= d->ldy, "conv fwd: zero_to > ldy");
  IgemmArgs a = {};
  a.g = fwd_geo(d, x);
  a.other = w; a.C = y; a.bias = bias;
  a.M = a.g.rows; a.N = d->Cout; a.ldc = d->ldy;
  if (d->stem4) { a.taps = d->kh; a.Cred = 32; a.ldo = d->kh * 32; a.tap_stride = 32; }
  else { a.taps = d->kh * d->kw; a.Cred = d->Cin; a.ldo = a.taps * d->Cin; a.tap_stride = d->Cin; }
  a.Cred_b = a.Cred; a.zero_to = zero_to; a.accumulate = 0;
  if (!d->stem4) fill_taps(a, d->kh, d->kw, d->dil, +1, d->stride, -d->pad);
  if (d->groups > 1) {  // one GEMM per group: batch strides walk the channel groups of x, w and y
    CS_REQUIRE(zero_to == 0, "conv fwd: zero_to is not supported with groups");
    const int cig = d->Cin / d->groups, cog = d->Cout / d->groups;
    a.N = cog; a.Cred = a.Cred_b = cig; a.ldo = a.taps * cig; a.tap_stride = cig;
    a.g_bs = cig; a.o_bs = (long long)cog * a.taps * cig; a.c_bs = cog;
    if (bias) { catseg_set_error("conv fwd: bias is not supported with groups"); return CATSEG_EINVAL; }
    return launch_igemm<L_NT>(a, d->groups, 1, (hipStream_t)stream);
  }
  return launch_igemm<L_NT>(a, 1, 1, (hipStream_t)stream);
}

// Training forward of conv -> BatchNorm: as catseg_conv2d_fwd, and the kernel's epilogue also writes the per-(M-tile, channel)
// BatchNorm partials for catseg_bn_finalize (no separate statistics pass over y).  *tile_rows / *n_tiles describe the
// partials ([n_tiles][3][Cout] floats in bn_part); *tile_rows == 0 means this layer's tile form has no fused statistics
// (grouped / 16-32 wide forms): the convolution has run, the caller uses catseg_bn_train_stats instead.
extern "C" int catseg_conv2d_fwd_bnstats(const catseg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                                         int zero_to, float* bn_part, size_t bn_part_floats, int* tile_rows, int* n_tiles,
                                         catseg_stream_t stream) {
  if (int e = check_desc(d)) return e;
  CS_REQUIRE(cs_aligned16(x) && cs_aligned16(w) && cs_aligned16(y), "conv fwd: pointers must be 16-byte aligned");
  CS_REQUIRE(zero_to <= d->ldy && tile_rows && n_tiles, "conv fwd bnstats: bad args");
  *tile_rows = 0; *n_tiles = 0;
  if (d->groups > 1) return catseg_conv2d_fwd(d, x, w, bias, y, zero_to, stream);
  IgemmArgs a = {};
  a.g = fwd_geo(d, x);
  a.other = w; a.C = y; a.bias = bias;
  a.M = a.g.rows; a.N = d->Cout; a.ldc = d->ldy;
  if (d->stem4) { a.taps = d->kh; a.Cred = 32; a.ldo = d->kh * 32; a.tap_stride = 32; }
  else { a.taps = d->kh * d->kw; a.Cred = d->Cin; a.ldo = a.taps * d->Cin; a.tap_stride = d->Cin; }
  a.Cred_b = a.Cred; a.zero_to = zero_to; a.accumulate = 0;
  if (!d->stem4) fill_taps(a, d->kh, d->kw, d->dil, +1, d->stride, -d->pad);
  const int ncols = zero_to > a.N ? zero_to : a.N;
  const TilePlan pl = plan_tiles(L_NT, a.M, ncols, 1, a.g.rows, true, (long long)a.taps * a.Cred);
  if (pl.narrow == 0 || pl.narrow == 1) {
    const int tm = 64 * pl.mi, nt = (a.M + tm - 1) / tm;
    if (bn_part != nullptr && (size_t)nt * 3 * d->Cout <= bn_part_floats) {
      a.bn_part = bn_part;
      *tile_rows = tm; *n_tiles = nt;
    }
  }
  return launch_igemm<L_NT>(a, 1, 1, (hipStream_t)stream, &pl);
}

// Inference: y = act(conv(x, w) + bias (+ residual)) in ONE kernel — Conv2d + eval-mode BatchNorm2d (folded into
// w / bias by catseg_fold_bn) + residual add + ReLU of a ResNet / ResNeXt / UPerNet block.
extern "C" int catseg_conv2d_fwd_fused(const catseg_conv_desc* d, const float* x, const float* w, const float* bias,
                                       const float* residual, int ldr, int relu, float* y, catseg_stream_t stream) {
  if (int e = check_desc(d)) return e;
  CS_REQUIRE(cs_aligned16(x) && cs_aligned16(w) && cs_aligned16(y), "conv fwd fused: pointers must be 16-byte aligned");
  IgemmArgs a = {};
  a.g = fwd_geo(d, x);
  a.other = w; a.C = y; a.bias = bias;
  a.M = a.g.rows; a.N = d->Cout; a.ldc = d->ldy;
  if (d->stem4) { a.taps = d->kh; a.Cred = 32; a.ldo = d->kh * 32; a.tap_stride = 32; }
  else { a.taps = d->kh * d->kw; a.Cred = d->Cin; a.ldo = a.taps * d->Cin; a.tap_stride = d->Cin; }
  a.Cred_b = a.Cred;
  a.residual = residual; a.ldr = ldr; a.relu = relu;
  if (!d->stem4) fill_taps(a, d->kh, d->kw, d->dil, +1, d->stride, -d->pad);
  if (d->groups > 1) {
    CS_REQUIRE(residual == nullptr, "conv fwd fused: residual is not supported with groups");
    const int cig = d->Cin / d->groups, cog = d->Cout / d->groups;
    a.N = cog; a.Cred = a.Cred_b = cig; a.ldo = a.taps * cig; a.tap_stride = cig;
    a.g_bs = cig; a.o_bs = (long long)cog * a.taps * cig; a.c_bs = cog;
    a.bias_bs = cog;
    return launch_igemm<L_NT>(a, d->groups, 1, (hipStream_t)stream);
  }
  return launch_igemm<L_NT>(a, 1, 1, (hipStream_t)stream);
}

extern "C" int catseg_conv2d_bwd_data(const catseg_conv_desc* d, const float* dy, const float* w, float* dx,
                                      int accumulate, catseg_stream_t stream) {
  if (int e = check_desc(d)) return e;
  CS_REQUIRE(!d->stem4, "conv bwd_data: not defined for the stem (the image needs no gradient)");
  CS_REQUIRE(cs_aligned16(dy) && cs_aligned16(w) && cs_aligned16(dx), "conv bwd_data: pointers must be 16-byte aligned");
  if (d->groups > 1) {
    // grouped (ResNeXt, models/ResNeXt.py:29-60 as a TRAINING encoder): one dense backward-data GEMM per group through the
    // batch strides -- group g reads dy channels [g cog, (g+1) cog), its filter bank, and writes dx channels [g cig, (g+1) cig)
    const int cig = d->Cin / d->groups, cog = d->Cout / d->groups;
    CS_REQUIRE(d->stride == 1 || d->kh * d->kw <= 32, "conv bwd_data: grouped strided convolution needs <= 32 taps");
    CS_REQUIRE(cog % 4 == 0, "conv bwd_data: Cout/groups must be a multiple of 4");
    catseg_conv_desc g1 = *d;
    g1.groups = 1; g1.Cin = cig; g1.Cout = cog;          // per-group geometry; ldx / ldy stay the full pixel strides
    // run the dense path once per group with shifted base pointers (stride > 1 launches several kernels per call)
    for (int g = 0; g < d->groups; ++g)
      if (int e = catseg_conv2d_bwd_data(&g1, dy + (size_t)g * cog, w + (size_t)g * cog * d->kh * d->kw * cig, dx + (size_t)g * cig, accumulate, stream))
        return e;
    return CATSEG_OK;
  }
  IgemmArgs a = {};
  Geo& g = a.g;
  g.base = dy; g.mode = 2; g.rows = d->B * d->H * d->W; g.ld = d->ldy;
  g.H = d->Ho; g.W = d->Wo; g.Ho = d->H; g.Wo = d->W; g.kw = d->kw; g.stride = d->stride; g.pad = d->pad; g.dil = d->dil;
  a.other = w; a.C = dx; a.bias = nullptr;
  a.M = g.rows; a.N = d->Cin; a.ldc = d->ldx;
  a.taps = d->kh * d->kw;
  a.Cred = (d->Cout + 3) & ~3;  // dy pad columns [Cout, roundup4) must be zero (conv_fwd zero_to / bilinear_bwd zero_to)
  a.Cred_b = d->Cout;
  a.ldo = a.taps * d->Cin; a.tap_stride = d->Cin;
  a.zero_to = 0; a.accumulate = accumulate;
  if (d->stride == 1) {
    fill_taps(a, d->kh, d->kw, d->dil, -1, 1, d->pad);
    return launch_igemm<L_NN>(a, 1, 1, (hipStream_t)stream);
  }
  // stride s > 1: one dense launch per parity class (py, px) of the input pixels.  Input row iy = a*s + py only
  // receives taps ky with (py + pad - ky*dil) % s == 0, from output row a + (py + pad - ky*dil) / s: no wasted MACs.
  const int sdv = d->stride;
  {
    // taps that reach ONE parity class (FCN's ConvTranspose2d 16x16 / stride 8, models/FCN.py:38: 2 x 2 of the 256); the tap table of a
    // launch holds 32
    int gd = sdv, v = d->dil;
    while (v) { const int t = gd % v; gd = v; v = t; }
    const int ks = sdv / gd;
    if (((d->kh + ks - 1) / ks) * ((d->kw + ks - 1) / ks) > 32) return launch_igemm<L_NN>(a, 1, 1, (hipStream_t)stream);   // generic (slow) path
  }
  IgemmArgs cls[4];
  int ncls = 0;
  const bool multi = sdv == 2 && g_strided_multi;
  for (int py = 0; py < sdv; ++py)
    for (int px = 0; px < sdv; ++px) {
      const int Hs = (d->H - py + sdv - 1) / sdv, Ws = (d->W - px + sdv - 1) / sdv;
      if (Hs <= 0 || Ws <= 0) continue;
      IgemmArgs q = a;
      q.g.Ho = Hs; q.g.Wo = Ws; q.g.rows = d->B * Hs * Ws; q.M = q.g.rows;
      q.row_s = 1; q.row_o = 0;
      // valid filter rows for this parity: ky = ky0 + i * kstep (kstep = s / gcd(s, dil)), i < nky
      auto gcd = [](int u, int v) { while (v) { const int w = u % v; u = v; v = w; } return u; };
      const int kstep = sdv / gcd(sdv, d->dil);
      int ky0 = -1, kx0 = -1;
      for (int k = 0; k < kstep && k < d->kh; ++k) if ((((py + d->pad - k * d->dil) % sdv) + sdv) % sdv == 0) { ky0 = k; break; }
      for (int k = 0; k < kstep && k < d->kw; ++k) if ((((px + d->pad - k * d->dil) % sdv) + sdv) % sdv == 0) { kx0 = k; break; }
      q.nky = ky0 < 0 ? 0 : (d->kh - ky0 + kstep - 1) / kstep;
      q.nkx = kx0 < 0 ? 0 : (d->kw - kx0 + kstep - 1) / kstep;
      if (q.nky > 0 && q.nkx > 0) {
        q.dy0 = (py + d->pad - ky0 * d->dil) / sdv; q.ddy = -(kstep * d->dil) / sdv;
        q.dx0 = (px + d->pad - kx0 * d->dil) / sdv; q.ddx = -(kstep * d->dil) / sdv;
        q.wy0 = ky0; q.wys = kstep; q.wx0 = kx0; q.wxs = kstep;
      } else {
        q.nky = q.nkx = 0;   // no tap reaches this parity class: the launch only writes zeros
      }
      q.taps = q.nky * q.nkx;
      q.remap = 1; q.out_s = sdv; q.out_py = py; q.out_px = px; q.out_H = d->H; q.out_W = d->W;
      if (multi) {
        int at = ncls++;
        while (at > 0 && cls[at - 1].taps < q.taps) { cls[at] = cls[at - 1]; --at; }   // heaviest class first
        cls[at] = q;
      } else if (int e = launch_igemm<L_NN>(q, 1, 1, (hipStream_t)stream)) {
        return e;
      }
    }
  if (multi && ncls > 0) return launch_multi_nn(cls, ncls, (hipStream_t)stream);
  return CATSEG_OK;
}

namespace {
TilePlan wgrad_plan(const catseg_conv_desc* d) {
  const long long rows = (long long)d->B * d->Ho * d->Wo;
  if (d->stem4) return plan_tiles(L_TN, d->Cout, 32, d->kh, rows);
  return plan_tiles(L_TN, d->Cout, (long long)d->kh * d->kw * d->Cin, 1, rows);
}
}  // namespace

extern "C" int catseg_debug_plan_conv(const catseg_conv_desc* d, int op, int* out) {
  if (int e = check_desc(d)) return e;
  CS_REQUIRE(out != nullptr && op >= 0 && op <= 2, "plan_conv: bad arguments");
  TilePlan pl;
  int direct = 0;
  if (op == 0) {
    const int g = d->groups > 1 ? d->groups : 1;
    pl = plan_tiles(L_NT, (long long)d->B * d->Ho * d->Wo, d->Cout / g, g, (long long)d->B * d->Ho * d->Wo, true, (long long)d->kh * d->kw * (d->Cin / g));
  } else if (op == 1) {
    const int s = d->stride;
    const long long rows = (long long)d->B * ((d->H + s - 1) / s) * ((d->W + s - 1) / s);
    pl = plan_tiles(L_NN, rows, d->Cin, (s == 2 && g_strided_multi) ? 4 : 1, rows);
  } else {
    pl = wgrad_plan(d);
    direct = wgrad_direct_workspace(d) > 0;
  }
  out[0] = pl.mi; out[1] = pl.ni; out[2] = pl.narrow; out[3] = pl.splits; out[4] = direct;
  return CATSEG_OK;
}

extern "C" size_t catseg_conv2d_bwd_weight_workspace(const catseg_conv_desc* d) {
  if (check_desc(d)) return 0;
  if (d->groups > 1) {
    catseg_conv_desc g1 = *d;
    g1.groups = 1; g1.Cin = d->Cin / d->groups; g1.Cout = d->Cout / d->groups;
    return catseg_conv2d_bwd_weight_workspace(&g1);
  }
  const int splits = wgrad_plan(d).splits;
  const size_t wel = (size_t)d->Cout * (d->stem4 ? d->kh * 32 : d->kh * d->kw * d->Cin);
  size_t bytes = splits > 1 ? (size_t)splits * wel * 4 : 0;
  const size_t direct = wgrad_direct_workspace(d);  // small-channel 3x3 layers take the direct kernel (wgrad_direct.hip)
  if (direct > bytes) bytes = direct;
  bytes += (size_t)256 * d->Cout * 4;  // bias-gradient partials
  return cs_align_up(bytes, 256);
}

extern "C" int catseg_conv2d_bwd_weight(const catseg_conv_desc* d, const float* x, const float* dy, float* dw,
                                        float* dbias, void* workspace, size_t workspace_bytes,
                                        catseg_stream_t stream) {
  if (int e = check_desc(d)) return e;
  CS_REQUIRE(cs_aligned16(dy) && cs_aligned16(x) && cs_aligned16(dw), "conv bwd_weight: pointers must be 16-byte aligned");
  if (d->groups > 1) {   // grouped: one dense backward-weight per group (shifted bases; the per-group filter banks are contiguous in dw)
    const int cig = d->Cin / d->groups, cog = d->Cout / d->groups;
    CS_REQUIRE(cog % 4 == 0 && cig % 4 == 0, "conv bwd_weight: channels per group must be multiples of 4");
    catseg_conv_desc g1 = *d;
    g1.groups = 1; g1.Cin = cig; g1.Cout = cog;
    for (int g = 0; g < d->groups; ++g)
      if (int e = catseg_conv2d_bwd_weight(&g1, x + (size_t)g * cig, dy + (size_t)g * cog, dw + (size_t)g * cog * d->kh * d->kw * cig,
                                           dbias ? dbias + (size_t)g * cog : nullptr, workspace, workspace_bytes, stream))
        return e;
    return CATSEG_OK;
  }
  const size_t need = catseg_conv2d_bwd_weight_workspace(d);
  if (workspace_bytes < need || (need && !workspace)) {
    catseg_set_error("conv bwd_weight: workspace %zu < %zu", workspace_bytes, need);
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const TilePlan pl = wgrad_plan(d);
  int splits = pl.splits;
  const int rps = pl.rps;
  const int taps = d->stem4 ? d->kh : d->kh * d->kw;
  const int ncol = d->stem4 ? 32 : d->Cin;
  const size_t wel = (size_t)d->Cout * taps * ncol;
  const int direct_slabs = wgrad_direct_launch(d, x, dy, workspace, st);
  if (direct_slabs > 0) {
    splits = direct_slabs;  // the slabs are reduced (and dbias computed) below exactly as for the split GEMM
    CS_LAUNCH_CHECK();
  }
  IgemmArgs a = {};
  a.g = fwd_geo(d, x);
  a.other = dy; a.ldo = d->ldy;
  a.M = d->Cout; a.ldc = taps * ncol;
  a.splits = splits; a.rows_per_split = rps; a.c_split_stride = (long long)wel;
  a.C = splits > 1 ? (float*)workspace : dw;
  a.taps = taps; a.Cred = 0; a.Cred_b = 0;
  const int grid_y = d->stem4 ? taps : 1;
  if (d->stem4) { a.N = ncol; a.c_tap_stride = ncol; a.tap_cin = 0; }
  else { a.N = taps * ncol; a.c_tap_stride = 0; a.tap_cin = ncol; }   // all taps side by side in the N dimension
  if (direct_slabs == 0)
    if (int e = launch_igemm<L_TN>(a, 1, grid_y, st, &pl)) return e;
  if (splits > 1) {
    const long long n4 = (long long)(wel / 4);
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    if (splits >= 64 && n4 <= 64 * 1024)  // many slabs of a small tensor: more parallelism across the slabs
      hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(1024), 0, st, (const float*)workspace, dw, n4, splits, n4);
    else
      hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n4, splits, n4);
    CS_LAUNCH_CHECK();
  }
  if (dbias) {
    float* part = (float*)((char*)workspace + (splits > 1 ? (size_t)splits * wel * 4 : 0));
    const long long rows = (long long)d->B * d->Ho * d->Wo;
    const int gy = (int)((rows + 1023) / 1024 < 256 ? (rows + 1023) / 1024 : 256);
    if (d->ldy == 32 && d->Cout <= 32 && cs_aligned16(dy)) {       // (the workspace holds 256 x Cout floats for the partials: 256 x 32 only then)
      const int gn = d->Cout == 32 ? gy : (int)((long long)gy * d->Cout / 32 > 0 ? (long long)gy * d->Cout / 32 : 1);
      hipLaunchKernelGGL(colsum_ld32_kernel, dim3(gn), dim3(256), 0, st, dy, rows, part);
      hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(256), 0, st, (const float*)part, gn, 32, dbias, d->Cout);
    } else {
      hipLaunchKernelGGL(colsum_partial_kernel, dim3((d->Cout + 63) / 64, gy), dim3(256), 0, st, dy, d->ldy, rows, d->Cout, part);
      hipLaunchKernelGGL(colsum_final_kernel, dim3((d->Cout + 255) / 256), dim3(256), 0, st, (const float*)part, gy, d->Cout, dbias);
    }
    CS_LAUNCH_CHECK();
  }
  return CATSEG_OK;
}

// dbias[o] = sum_p dy[p][o] on its own (the split-precision backward-weight path computes only dW); workspace >= 256 * C floats
extern "C" int catseg_bias_grad(const float* dy, int ld, long long rows, int C, float* dbias, void* workspace, size_t workspace_bytes,
                                catseg_stream_t stream) {
  CS_REQUIRE(dy && dbias && rows > 0 && C > 0 && ld >= C, "bias_grad: bad args");
  CS_REQUIRE(workspace && workspace_bytes >= (size_t)256 * C * 4, "bias_grad: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int gy = (int)((rows + 1023) / 1024 < 256 ? (rows + 1023) / 1024 : 256);
  if (ld == 32 && C <= 32 && cs_aligned16(dy) && workspace_bytes >= (size_t)256 * 32 * 4) {
    hipLaunchKernelGGL(colsum_ld32_kernel, dim3(gy), dim3(256), 0, st, dy, rows, (float*)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, gy, 32, dbias, C);
    CS_LAUNCH_CHECK();
    return CATSEG_OK;
  }
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, gy), dim3(256), 0, st, dy, ld, rows, C, (float*)workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)workspace, gy, C, dbias);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_gemm_batched(int layout, int batch, int M, int N, int K, const float* A, int lda,
                                   long long strideA, const float* Bm, int ldb, long long strideB, float* C,
                                   int ldc, long long strideC, int zero_to, int accumulate,
                                   catseg_stream_t stream) {
  CS_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0, "gemm: bad dims");
  CS_REQUIRE(cs_aligned16(A) && cs_aligned16(Bm) && cs_aligned16(C), "gemm: pointers must be 16-byte aligned");
  CS_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && strideA % 4 == 0 && strideB % 4 == 0, "gemm: lda/ldb/strides must be multiples of 4");
  CS_REQUIRE(zero_to <= ldc, "gemm: zero_to > ldc");
  CS_REQUIRE((long long)M * lda < (1ll << 31) && (long long)N * ldb < (1ll << 31) && (long long)K * (lda > ldb ? lda : ldb) < (1ll << 31),
             "gemm: operands must have fewer than 2^31 elements per batch item");
  IgemmArgs a = {};
  Geo& g = a.g;
  g.mode = 1; g.W = g.Wo = 1; g.kw = 1; g.stride = 1; g.pad = 0; g.dil = 1;  // rows = a 1-pixel-wide image
  a.C = C; a.ldc = ldc; a.c_bs = strideC; a.M = M; a.N = N; a.zero_to = zero_to; a.accumulate = accumulate;
  a.taps = 1; a.tap_stride = 0;
  a.row_s = 1; a.row_o = 0; a.nky = a.nkx = 1; a.dy0 = a.ddy = a.dx0 = a.ddx = 0; a.wy0 = a.wx0 = 0; a.wys = a.wxs = 1;
  hipStream_t st = (hipStream_t)stream;
  if (layout == CATSEG_GEMM_NT) {
    CS_REQUIRE(K % 4 == 0 && lda >= K && ldb >= K, "gemm NT: K must be a multiple of 4 and <= lda, ldb");
    g.base = A; g.rows = M; g.H = g.Ho = M; g.ld = lda; a.g_bs = strideA;
    a.other = Bm; a.ldo = ldb; a.o_bs = strideB; a.Cred = K; a.Cred_b = K;
    return launch_igemm<L_NT>(a, batch, 1, st);
  } else if (layout == CATSEG_GEMM_NN) {
    CS_REQUIRE(lda >= ((K + 3) & ~3) && ldb >= ((N + 3) & ~3), "gemm NN: lda/ldb too small");
    g.base = A; g.rows = M; g.H = g.Ho = M; g.ld = lda; a.g_bs = strideA;
    a.other = Bm; a.ldo = ldb; a.o_bs = strideB; a.Cred = (K + 3) & ~3; a.Cred_b = K;
    return launch_igemm<L_NN>(a, batch, 1, st);
  } else if (layout == CATSEG_GEMM_TN) {
    CS_REQUIRE(lda >= ((M + 3) & ~3) && ldb >= ((N + 3) & ~3), "gemm TN: lda/ldb too small");
    g.base = Bm; g.rows = K; g.H = g.Ho = K; g.ld = ldb; a.g_bs = strideB;
    a.other = A; a.ldo = lda; a.o_bs = strideA;
    a.splits = 1; a.rows_per_split = (K + 15) / 16 * 16; a.c_split_stride = 0; a.c_tap_stride = 0;
    a.tap_cin = (N + 3) & ~3;
    TilePlan pl = plan_tiles(L_NT, M, zero_to > N ? zero_to : N, batch, K, false);  // no split-K for the batched form
    pl.splits = 1;
    return launch_igemm<L_TN>(a, batch, 1, st, &pl);
  }
  catseg_set_error("gemm: unknown layout %d", layout);
  return CATSEG_EINVAL;
}
