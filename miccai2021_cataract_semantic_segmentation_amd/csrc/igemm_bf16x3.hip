// Split-precision implicit-GEMM convolution on the bf16 matrix cores of gfx950.
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  An fp32 value splits EXACTLY into three bf16 pieces
//   x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); 3 x 8 significant bits + signs cover the 24-bit
// significand, residual <= 2^-25 |x|), and a product of two such sums is reproduced to ~2^-23 relative by the six partial
// products of order <= 2 (hh, hm, mh, hl, lh, mm); each bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32.
// Six v_mfma_f32_32x32x16_bf16 per 32x32x16 block instead of eight v_mfma_f32_32x32x2_f32 at 1/16 of the rate:
// 16 / 6 = 2.7x the fp32-matrix peak (157 TFLOP/s -> 419 TFLOP/s-equivalent).  Results are NOT bit-identical to an fp32 FMA
// chain (the summation order and the dropped third-order terms differ at the 1e-7 level) -- see DESIGN.md for the measured error.
//
// Operands are pre-split planes written by split3_kernel: plane p of an [rows][C] fp32 tensor is a [rows][ldp] bf16 array
// (ldp = C rounded up to 8, rows 16-byte aligned); conv weights OHWI -> [3][O][taps * Cin].
//
// Kernel: 512 threads (8 waves, 2 per SIMD, one block per CU), block tile (32 TM WGM) x (32 TN WGN), K-step = 16 bf16 =
// one MFMA deep; LDS image per operand plane: [rows][32 B], the two 16-byte chunks of a row swapped where (row >> 3) & 1
// (conflict-free ds_read_b128 over its four 16-lane groups); global -> LDS by global_load_lds_dwordx4, double buffered
// behind a counted vmcnt.  The im2col matrix is never built (per-row tap mask + affine tap offsets, zero page for padding).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __attribute__((aligned(256))) float g_zero_page_b3[64];

struct B3Args {
  const u16* a;       // activation planes
  long long a_plane;  // elements between planes
  int lda;            // elements per pixel row
  const u16* w;       // weight planes [N][ldw]
  long long w_plane;
  int ldw;            // = taps * Cin
  float* C;
  int ldc;
  const float* bias;
  int M, N, Cin, taps;
  int H, W, Ho, Wo;   // H, W: source image of the gather; Ho, Wo: the grid the GEMM rows decode over
  int kw, stride, pad, dil;
  int sign;           // +1: forward gather (iy = y*stride - pad + ky*dil); -1: stride-1 backward-data gather (iy = y + pad - ky*dil)
  int tilesM, tilesN;
  int zero_to, accumulate;
  const float* zero;
  float* bn_part;     // per (M-tile, channel) BatchNorm partials [tilesM][3][N], or nullptr
  // blocked operand planes (igemm_b3w_kernel<true> only): activations [Cin/16][a_rows][16], weights [taps*Cin/16][w_rows][16]
  int blocked, a_rows, w_rows;
  // fused inference epilogue (igemm_b3w_kernel): v = act(v + bias (+ residual))
  const float* residual;
  int ldr, relu;
};

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// fp32 -> three bf16 planes.  One thread = 8 consecutive channels of one row.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int ld, long long rows, int C, int ldp,
                                                     u16* __restrict__ out, long long plane) {
  const int c8 = ldp >> 3;
  const long long n = rows * c8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c8;
    const int c0 = (int)(i - r * c8) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c0 + j < C) ? x[r * ld + c0 + j] : 0.f;
    bf16x8 h, m, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 hh = (__bf16)v[j];
      const float r1 = v[j] - (float)hh;
      const __bf16 mm = (__bf16)r1;
      const float r2 = r1 - (float)mm;
      h[j] = hh; m[j] = mm; l[j] = (__bf16)r2;
    }
    bf16x8* o = (bf16x8*)(out + r * ldp + c0);
    *o = h;
    *(bf16x8*)((u16*)o + plane) = m;
    *(bf16x8*)((u16*)o + 2 * plane) = l;
  }
}

// OHWI fp32 weights -> planes of the TRANSPOSED filter bank [Cin][taps][O] (the B operand of backward-data as an NT GEMM)
__global__ __launch_bounds__(256) void split3_wt_kernel(const float* __restrict__ w, int O, int taps, int Cin, int ldp,
                                                        u16* __restrict__ out, long long plane) {
  const long long n = (long long)Cin * taps * ldp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(i % ldp);
    const long long ct = i / ldp;
    const int t = (int)(ct % taps), c = (int)(ct / taps);
    const float v = o < O ? w[((long long)o * taps + t) * Cin + c] : 0.f;
    const __bf16 hh = (__bf16)v;
    const float r1 = v - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const __bf16 ll = (__bf16)(r1 - (float)mm);
    out[i] = __builtin_bit_cast(u16, hh);
    out[i + plane] = __builtin_bit_cast(u16, mm);
    out[i + 2 * plane] = __builtin_bit_cast(u16, ll);
  }
}

__device__ __forceinline__ void split3_one(float v, u16& h, u16& m, u16& l) {
  const __bf16 hh = (__bf16)v;
  const float r1 = v - (float)hh;
  const __bf16 mm = (__bf16)r1;
  h = __builtin_bit_cast(u16, hh);
  m = __builtin_bit_cast(u16, mm);
  l = __builtin_bit_cast(u16, (__bf16)(r1 - (float)mm));
}

// fp32 [rows][ld] -> three bf16 planes in the BLOCKED layout [C16][rows][16] (C16 = ceil(C / 16), channel tail zero) and, in the
// same pass, optionally the planar layout [rows][ldp] of split3_kernel.  Block = 64 rows x 128 channels through LDS: the source
// is read with 512-byte row segments, the blocked planes are written as 2 KB runs (64 rows x 32 B of one chunk), the planar planes
// as 256-byte row segments.
__global__ __launch_bounds__(256) void split3_blocked_kernel(const float* __restrict__ x, int ld, long long rows, int C, int ldp,
                                                             u16* __restrict__ blk, long long blk_plane, u16* __restrict__ planar,
                                                             long long planar_plane) {
  constexpr int RS = 128 + 8;                      // LDS row stride in bf16 elements (16-byte aligned, skewed)
  __shared__ __attribute__((aligned(16))) u16 sh[3][64 * RS];
  const long long r0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 128;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = i * 256 + t, row = q >> 5, c4 = (q & 31) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r0 + row < rows) {
      const float* src = x + (r0 + row) * ld + c0 + c4;
      if (c0 + c4 + 3 < C) {
        const f32x4 f = *(const f32x4*)src;
        v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (c0 + c4 + j < C) ? src[j] : 0.f;
      }
    }
    u16 h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3_one(v[j], h[j], m[j], l[j]);
    u16* d = &sh[0][row * RS + c4];
    *(unsigned long long*)d = (unsigned long long)h[0] | ((unsigned long long)h[1] << 16) | ((unsigned long long)h[2] << 32) | ((unsigned long long)h[3] << 48);
    *(unsigned long long*)(d + 64 * RS) = (unsigned long long)m[0] | ((unsigned long long)m[1] << 16) | ((unsigned long long)m[2] << 32) | ((unsigned long long)m[3] << 48);
    *(unsigned long long*)(d + 2 * 64 * RS) = (unsigned long long)l[0] | ((unsigned long long)l[1] << 16) | ((unsigned long long)l[2] << 32) | ((unsigned long long)l[3] << 48);
  }
  __syncthreads();
  if (planar != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + t, row = q >> 4, c8 = (q & 15) * 8;
      if (r0 + row < rows && c0 + c8 < ldp) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          *(bf16x8*)(planar + pl * planar_plane + (r0 + row) * ldp + c0 + c8) = *(const bf16x8*)&sh[pl][row * RS + c8];
      }
    }
  }
  const int c16 = (C + 15) >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + t, cc = q >> 7, row = (q & 127) >> 1, half = (q & 1) * 8;
    const int chunk = (c0 >> 4) + cc;
    if (r0 + row < rows && chunk < c16) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        *(bf16x8*)(blk + pl * blk_plane + ((long long)chunk * rows + r0 + row) * 16 + half) = *(const bf16x8*)&sh[pl][row * RS + cc * 16 + half];
    }
  }
}

// fp32 weights viewed as [N][K] (row n, reduction index k; source element = w[n * sn + k * sk]) -> blocked planes [K/16][N][16]
// (K a multiple of 16): one K-step's 256 tile rows are one contiguous run.  T = the transposed filter bank of backward-data:
// row n = input channel c, k = tap * Opad + o, source w[(o * taps + tap) * Cin + c], zero for o >= O.
template <bool T>
__global__ __launch_bounds__(256) void split3_weight_blocked_kernel(const float* __restrict__ w, int N, int K, int O, int Opad, int taps, int Cin,
                                                                    u16* __restrict__ out, long long plane) {
  const long long n_el = (long long)N * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 15);
    const long long rest = i >> 4;
    const int n = (int)(rest % N), k16 = (int)(rest / N);
    const int k = k16 * 16 + kk;
    float v;
    if (T) {
      const int tap = k / Opad, o = k - tap * Opad;
      v = o < O ? w[((long long)o * taps + tap) * Cin + n] : 0.f;
    } else {
      v = w[(long long)n * K + k];
    }
    u16 h, m, l;
    split3_one(v, h, m, l);
    out[i] = h;
    out[i + plane] = m;
    out[i + 2 * plane] = l;
  }
}

template <int TM, int TN, int WGM, int WGN, int NBUF>
__global__ __launch_bounds__(512, 2) void igemm_b3_kernel(const B3Args p) {
  static_assert(WGM * WGN == 8, "8 waves");
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  constexpr int PLANE_A = BM * 32, PLANE_B = BN * 32;     // bytes of one plane image (rows x 32 B)
  constexpr int SLAB = 3 * (PLANE_A + PLANE_B);           // bytes per K-step buffer
  constexpr int NA = (BM * 2 + 511) / 512, NB = (BN * 2 + 511) / 512;   // 16-byte chunks per thread per plane
  __shared__ __attribute__((aligned(16))) char smem[NBUF * SLAB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, h = lane >> 5;

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = swz / p.tilesN, tile_n = swz - tile_m * p.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- staging state: chunk q = j*512 + tid of a plane image: row = q >> 1, position = q & 1 (lane-linear LDS-DMA),
  //      logical 8-channel chunk = position ^ ((row >> 3) & 1)
  int aoff[NA];
  unsigned amask[NA];
  int achunk[NA];
  bool alive[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int q = j * 512 + tid, row = q >> 1;
    alive[j] = row < BM;
    achunk[j] = (q & 1) ^ ((row >> 3) & 1);
    const int r = m0 + row;
    aoff[j] = 0;
    amask[j] = 0;
    if (alive[j] && r < p.M) {
      const int hw = p.Ho * p.Wo;
      const int b = r / hw, rem = r - b * hw;
      const int y = rem / p.Wo, x = rem - y * p.Wo;
      const int y0 = p.sign > 0 ? y * p.stride - p.pad : y + p.pad;
      const int x0 = p.sign > 0 ? x * p.stride - p.pad : x + p.pad;
      aoff[j] = ((b * p.H + y0) * p.W + x0) * p.lda;
      const int kh = p.taps / p.kw;
      int t = 0;
      for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < p.kw; ++kx, ++t) {
          const int yy = y0 + p.sign * ky * p.dil, xx = x0 + p.sign * kx * p.dil;
          if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) amask[j] |= 1u << t;
        }
    }
  }
  int bchunk[NB], brow[NB];
  bool blive[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int q = j * 512 + tid, row = q >> 1;
    blive[j] = row < BN;
    bchunk[j] = (q & 1) ^ ((row >> 3) & 1);
    brow[j] = n0 + row;
  }

  const int nck = (p.Cin + 15) >> 4;
  const int nks = p.taps * nck;
  int ttap = 0, tky = 0, tkx = 0, tck = 0;
  const u16* pa[NA];
  const u16* pb[NB];
  const u16* zero = (const u16*)p.zero;

  auto prep = [&]() {
    const int toff = p.sign * (tky * p.dil * p.W + tkx * p.dil) * p.lda + tck * 16;
    const int woff = ttap * p.Cin + tck * 16;
    const bool tap_ok = ttap < p.taps;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int c = achunk[j] * 8;
      const bool ok = tap_ok && ((amask[j] >> (ttap & 31)) & 1u) && (tck * 16 + c) < p.Cin;
#ifdef B3G_BLOCKED   // address-only timing experiment (tools/ab_b3w.sh): activation planes [C/16][pixel][16]; values are garbage
      pa[j] = ok ? p.a + ((long long)(tck * p.M + aoff[j] / p.lda + p.sign * (tky * p.dil * p.W + tkx * p.dil)) * 16 + (c & 8)) : zero;
#else
      pa[j] = ok ? p.a + (aoff[j] + toff + c) : zero;
#endif
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int c = bchunk[j] * 8;
      const bool ok = tap_ok && brow[j] < p.N && (tck * 16 + c) < p.Cin;
#ifdef B3G_BLOCKED   // weight planes [K/16][N][16]
      pb[j] = ok ? p.w + ((long long)((ttap * nck + tck) * p.N + brow[j]) * 16 + (c & 8)) : zero;
#else
      pb[j] = ok ? p.w + ((long long)brow[j] * p.ldw + woff + c) : zero;
#endif
    }
    if (++tck == nck) {
      tck = 0;
      ++ttap;
      if (++tkx == p.kw) {
        tkx = 0;
        ++tky;
      }
    }
  };

  auto issue = [&](const int buf) {
    char* s = smem + buf * SLAB;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int j = 0; j < NA; ++j)
        if (alive[j]) glds16(pa[j] == zero ? zero : pa[j] + pl * p.a_plane, s + pl * PLANE_A + (j * 512 + wave * 64) * 16);
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (blive[j]) glds16(pb[j] == zero ? zero : pb[j] + pl * p.w_plane, s + 3 * PLANE_A + pl * PLANE_B + (j * 512 + wave * 64) * 16);
    }
  };

  auto compute = [&](const int buf, const bool more) {
    const char* s = smem + buf * SLAB;
    bf16x8 a[3][TM], b[3][TN];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = wm * 32 * TM + t * 32 + l31;
        a[pl][t] = *(const bf16x8*)(s + pl * PLANE_A + row * 32 + ((h ^ ((row >> 3) & 1)) << 4));
      }
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const int row = wn * 32 * TN + u * 32 + l31;
        b[pl][u] = *(const bf16x8*)(s + 3 * PLANE_A + pl * PLANE_B + row * 32 + ((h ^ ((row >> 3) & 1)) << 4));
      }
    }
    if (more) prep();
    // six partial products in the order the planes were read (h first): the MFMAs on the h planes start while the m / l
    // fragment reads are still in flight (counted lgkmcnt); the accumulation order is irrelevant at this precision
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][t], b[PB[q]][u], acc[t][u], 0, 0, 0);
  };

  // LDS-DMA instructions THIS wave issues per K-step (wave-uniform: a wave covers 32 whole rows of a plane image; waves
  // beyond the end of a narrow operand issue fewer) -- the counted vmcnt must match it exactly
  int nload = 0;
#pragma unroll
  for (int j = 0; j < NA; ++j) nload += (j * 512 + wave * 64) < BM * 2 ? 3 : 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) nload += (j * 512 + wave * 64) < BN * 2 ? 3 : 0;
  nload = __builtin_amdgcn_readfirstlane(nload);
  auto wait_one_step = [&]() {   // all but the youngest K-step's loads of this wave have landed
    switch (nload) {
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  if (nks > 0 && NBUF == 2) {
    prep();
    issue(0);
    prep();
    for (int ks = 0; ks < nks; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < nks) {
        issue(cur ^ 1);
        wait_one_step();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      compute(cur, ks + 2 < nks);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else if (nks > 0) {
    // three-slot ring, ONE barrier per K-step: K-steps k+1 and k+2 are in flight while k is computed.  The slot refilled after
    // the barrier of step k was last read in step k-1, which every wave has left (its fragment reads are complete: lgkmcnt(0)
    // before it entered the barrier).
    prep();
    issue(0);
    prep();
    if (nks > 1) issue(1);
    prep();
    int cur = 0, fill = 2;
    for (int ks = 0; ks < nks; ++ks) {
      if (ks + 1 < nks) wait_one_step();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (ks + 2 < nks) issue(fill);
      compute(cur, ks + 3 < nks);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      cur = cur == 2 ? 0 : cur + 1;
      fill = fill == 2 ? 0 : fill + 1;
    }
  }

  if (p.bn_part != nullptr) {   // BatchNorm batch statistics of this tile (common.h: cs_tile_bn_partials)
    __syncthreads();
    int colv[TN];
    float bv[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      colv[u] = wn * 32 * TN + u * 32 + l31;
      bv[u] = (p.bias != nullptr && n0 + colv[u] < p.N) ? p.bias[n0 + colv[u]] : 0.f;
    }
    const int rbase = m0 + wm * 32 * TM + 4 * h;
    cs_tile_bn_partials<TN, TM * 16, WGM, false>(
        (float*)smem, BN, colv, h == 0, wm, min(BM, p.M - m0),
        [&](int j, int i) { return acc[i >> 4][j][i & 15] + bv[j]; },
        [&](int i) { return rbase + (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2) < p.M; }, p.bn_part + (long long)tile_m * 3 * p.N, p.N, n0);
  }
  // ---- epilogue (C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int col = n0 + wn * 32 * TN + u * 32 + l31;
      const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 * TM + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < p.M) {
          float* dst = p.C + (long long)row * p.ldc + col;
          if (col < p.N) {
            float v = acc[t][u][r] + bv;
            if (p.accumulate) v += *dst;
            *dst = v;
          } else if (col < p.zero_to) {
            *dst = 0.f;
          }
        }
      }
    }
}


// ---- the large-tile form: 4 waves (ONE per SIMD, up to 512 registers each), block tile 256 x 256, wave tile 128 x 128.
// With one wave per SIMD nothing else can fill the matrix pipe while a wave waits, so every memory operation is software-
// pipelined under the wave's own MFMA stream (96 MFMAs = 3072 cycles per K-step).  The six products run in the order
//   hh | hl lh | hm mh | mm
// and the fragments of the NEXT K-step are re-read into the same registers right after a plane's last use: l after "hl lh",
// h after "hm mh", m after "mm"; each is needed 16 ... 64 MFMAs (512 ... 2048 cycles) later, so the matrix pipe never waits
// for LDS and no fragment is double buffered (96 fragment registers + 256 accumulators in AGPRs).
// LDS: three 48 KB slots.  K-step k computes from registers, reads the fragments of k+1 from slot (k+1) % 3, while the LDS-DMA
// of k+2 (issued right after the barrier at the top of step k) fills slot (k+2) % 3, whose last reader was step k-1.
// ONE barrier per K-step; the loads have a whole K-step of latency budget.

// s_waitcnt immediate of gfx9 (vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]); issued through the builtin so that the
// compiler's own wait-count insertion knows the counters' state (inline asm is opaque to it)
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

#ifndef B3X_TAPS_INNER
#define B3X_TAPS_INNER 1
#endif
template <bool BLK>
__global__ __launch_bounds__(256, 1) void igemm_b3w_kernel(const B3Args p) {
  constexpr int TM = 4, TN = 4, WGN = 2;
  constexpr int BM = 256, BN = 256;
  constexpr int PLANE_A = BM * 32, PLANE_B = BN * 32;
  constexpr int SLAB = 3 * (PLANE_A + PLANE_B);
  constexpr int NA = 2, NB = 2;   // 16-byte chunks per thread per plane
  __shared__ __attribute__((aligned(16))) char smem[3 * SLAB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, h = lane >> 5;

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = swz / p.tilesN, tile_n = swz - tile_m * p.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int aoff[NA], achunk[NA], bchunk[NB], brow[NB];
  unsigned amask[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int q = j * 256 + tid, row = q >> 1;
    achunk[j] = ((q & 1) ^ ((row >> 3) & 1)) * 8;
    const int r = m0 + row;
    aoff[j] = 0;
    amask[j] = 0;
    if (r < p.M) {
      const int hw = p.Ho * p.Wo;
      const int b = r / hw, rem = r - b * hw;
      const int y = rem / p.Wo, x = rem - y * p.Wo;
      const int y0 = p.sign > 0 ? y * p.stride - p.pad : y + p.pad;
      const int x0 = p.sign > 0 ? x * p.stride - p.pad : x + p.pad;
      aoff[j] = BLK ? (b * p.H + y0) * p.W + x0 : ((b * p.H + y0) * p.W + x0) * p.lda + achunk[j];   // BLK: pixel index of tap (0, 0)
      const int kh = p.taps / p.kw;
      int t = 0;
      for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < p.kw; ++kx, ++t) {
          const int yy = y0 + p.sign * ky * p.dil, xx = x0 + p.sign * kx * p.dil;
          if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) amask[j] |= 1u << t;
        }
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int q = j * 256 + tid, row = q >> 1;
    bchunk[j] = ((q & 1) ^ ((row >> 3) & 1)) * 8;
    brow[j] = n0 + row;
  }
  int boff[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) boff[j] = brow[j] < p.N ? brow[j] * p.ldw + bchunk[j] : -1;   // (weights < 2^31 elements: checked on the host)

  const int nck = (p.Cin + 15) >> 4;
  const int nks = p.taps * nck;
  int ttap = 0, tky = 0, tkx = 0, tck = 0;
  // LDS-DMA source addressing through two raw buffer resources (one per operand, all three planes inside; the host takes this
  // kernel only when 3 planes < 4 GB): a lane's source = 32-bit BYTE offset; padding taps, channel tails and rows past M / N get an
  // out-of-range offset, for which the hardware writes zeros into LDS (tools/probe/buffer_lds_probe.hip) -- no zero page, no
  // 64-bit pointer arithmetic: per K-step 4 offsets instead of 12 pointers.
  const unsigned apl_b = (unsigned)(p.a_plane * 2), wpl_b = (unsigned)(p.w_plane * 2);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, (short)0, (int)(3u * apl_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)(3u * wpl_b), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned VA[NA], VB[NB];   // byte offsets of one K-step's pieces inside plane 0 (the plane offset rides in the instruction's soffset)
  auto prepA = [&](const int j) {
    const int toff = p.sign * (tky * p.dil * p.W + tkx * p.dil) * p.lda + tck * 16;
    const bool ok = (int)(ttap < p.taps) & (int)((amask[j] >> (ttap & 31)) & 1u) & (int)((tck * 16 + achunk[j]) < p.Cin);
    // BLK: planes laid out [Cin/16][pixel][16]: the 256 rows of a K-step are one contiguous run of 32-byte pieces (whole
    // cache lines per LDS-DMA instruction; with [pixel][Cin] planes every lane pair touched its own line: the vector L1's line
    // rate, not HBM or L2, bounded the kernel at 190 TFLOP/s-equivalent -- 248 with both operands blocked)
    const unsigned v = BLK ? (unsigned)((tck * p.a_rows + aoff[j] + p.sign * (tky * p.dil * p.W + tkx * p.dil)) * 16 + achunk[j]) * 2u
                           : (unsigned)(aoff[j] + toff) * 2u;
    VA[j] = ok ? v : OOB;
  };
  auto prepB = [&](const int j) {
    const int woff = ttap * p.Cin + tck * 16;
    const bool ok = (int)(ttap < p.taps) & (int)(boff[j] >= 0) & (int)((tck * 16 + bchunk[j]) < p.Cin);
    const unsigned v = BLK ? (unsigned)(((ttap * nck + tck) * p.w_rows + brow[j]) * 16 + bchunk[j]) * 2u   // [K/16][N][16]
                           : (unsigned)(boff[j] + woff) * 2u;
    VB[j] = ok ? v : OOB;
  };
  // K order.  BLK: 16-channel chunk outer, filter taps inner -- the nine taps of a chunk re-read the same contiguous run of a blocked
  // plane shifted by a few pixels (L1 / L2 hits instead of nine passes over the whole plane through the fabric); planar planes keep
  // taps outer, chunks inner (consecutive K-steps walk along a pixel row's cache lines).
  auto advance = [&]() {
    if (BLK && B3X_TAPS_INNER) {
      const int nx = tkx + 1, nt = ttap + 1;
      const bool wrapx = nx == p.kw, wrapt = nt == p.taps;
      tkx = wrapx ? 0 : nx;
      tky = wrapt ? 0 : (wrapx ? tky + 1 : tky);
      ttap = wrapt ? 0 : nt;
      tck += wrapt ? 1 : 0;
    } else {
      const int nt = tck + 1, nx = tkx + 1;
      const bool wrap = nt == nck, wrapx = wrap && (nx == p.kw);
      tck = wrap ? 0 : nt;
      ttap += wrap ? 1 : 0;
      tkx = wrap ? (wrapx ? 0 : nx) : tkx;
      tky += wrapx ? 1 : 0;
    }
  };
  auto prep = [&]() {
    prepA(0); prepA(1); prepB(0); prepB(1); advance();
  };
  // LDS-DMA piece i of a K-step: i = 4 * plane + {A0, A1, B0, B1}
  auto piece = [&](const int buf, const int i) {
    char* s = smem + buf * SLAB;
    const int pl = i >> 2, w = i & 3;
    if (w < 2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(s + pl * PLANE_A + (w * 256 + wave * 64) * 16), 16,
                                               VA[w], pl * apl_b, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(s + 3 * PLANE_A + pl * PLANE_B + ((w - 2) * 256 + wave * 64) * 16),
                                               16, VB[w - 2], pl * wpl_b, 0, 0);
  };
  auto issue = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < 12; ++i) piece(buf, i);
  };

  bf16x8 Ah[TM], Bh[TN], Al[TM], Bl[TN], Am[TM], Bm[TN];
  // Fragment reads are ordinary LDS loads (the compiler inserts the waits; at the loop header it is conservative and waits
  // for the five trailing m' reads issued just before, which have had the barrier's time to complete).  (Inline-asm reads with
  // hand-counted lgkmcnt waits were tried: the register allocator copies asm results before they have arrived.)
  int ra0, rb0;
  {
    const int rowa = wm * 128 + l31, rowb = wn * 128 + l31;   // (row + 32 t keeps (row >> 3) & 1: one swizzle per lane)
    ra0 = rowa * 32 + ((h ^ ((rowa >> 3) & 1)) << 4);
    rb0 = 3 * PLANE_A + rowb * 32 + ((h ^ ((rowb >> 3) & 1)) << 4);
  }
// (differential timing builds, tools/ab_b3w.sh: -DB3X_NO_PREP / _NO_DMA / _NO_READS / _NO_SYNC drop one ingredient of the K loop -- wrong
//  results, the time difference is that ingredient's cost)
#ifdef B3X_NO_DMA
#define B3X_PIECE(f, i) do {} while (0)
#else
#define B3X_PIECE(f, i) piece(f, i)
#endif
#ifdef B3X_NO_PREP
#define B3X_PREPA(j) do {} while (0)
#define B3X_PREPB(j) do {} while (0)
#define B3X_ADVANCE() do {} while (0)
#else
#define B3X_PREPA(j) prepA(j)
#define B3X_PREPB(j) prepB(j)
#define B3X_ADVANCE() advance()
#endif
#ifdef B3X_NO_READS
#define B3X_READ(dst, base, off) do {} while (0)
#else
#define B3X_READ(dst, base, off) B3_DS_READ(dst, base, off)
#endif
#ifdef B3X_NO_SYNC
#define B3X_SYNC() do {} while (0)
#else
#define B3X_SYNC()                                        \
  do {                                                    \
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(12, 0));       \
    __builtin_amdgcn_s_barrier();                         \
  } while (0)
#endif
#define B3_DS_READ(dst, base, off) dst = *(const bf16x8*)((base) + (off))
#define B3_READ_PLANE(slot, pl, A_, B_)                                                \
  do {                                                                                 \
    const char* aa_ = smem + (slot) * SLAB + ra0;                                      \
    const char* bb_ = smem + (slot) * SLAB + rb0;                                      \
    B3_DS_READ(A_[0], aa_, (pl) * PLANE_A + 0 * 1024);                                 \
    B3_DS_READ(A_[1], aa_, (pl) * PLANE_A + 1 * 1024);                                 \
    B3_DS_READ(A_[2], aa_, (pl) * PLANE_A + 2 * 1024);                                 \
    B3_DS_READ(A_[3], aa_, (pl) * PLANE_A + 3 * 1024);                                 \
    B3_DS_READ(B_[0], bb_, (pl) * PLANE_B + 0 * 1024);                                 \
    B3_DS_READ(B_[1], bb_, (pl) * PLANE_B + 1 * 1024);                                 \
    B3_DS_READ(B_[2], bb_, (pl) * PLANE_B + 2 * 1024);                                 \
    B3_DS_READ(B_[3], bb_, (pl) * PLANE_B + 3 * 1024);                                 \
  } while (0)
#define B3_MFMA(A_, t_, B_, u_)                                                                                \
  do {                                                                                                        \
    acc[t_][u_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[t_], B_[u_], acc[t_][u_], 0, 0, 0);               \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)
  auto mm = [&](const bf16x8* a, const bf16x8* b) {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[u], acc[t][u], 0, 0, 0);
  };

  if (nks > 0) {
    prep();
    issue(0);
    prep();
    issue(1);                       // (unconditional: past the end of the reduction every offset is out of range = zeros)
    prep();
    issue(2);
    prep();
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    B3_READ_PLANE(0, 2, Al, Bl);
    B3_READ_PLANE(0, 0, Ah, Bh);
    B3_READ_PLANE(0, 1, Am, Bm);
    // The K loop is rotated: an iteration = the second half of K-step k (products hm, mh, mm: they retire every fragment set, so
    // the h / l fragment reads of step k+1 and the 12 LDS-DMA pieces of step k+2 ride there) followed by the first half of step
    // k+1 (hh, hl, lh + the address arithmetic of step k+3; the m-plane fragments, retired by mm, are read here: nothing in this half
    // waits for them).  The loop header -- where the compiler waits for ALL outstanding LDS reads -- and the barrier then sit
    // behind ~40 MFMAs without a new read.  One wave per SIMD issues in order, so every MFMA carries at most one other
    // operation, pinned by scheduling fences.  Step nks is all zeros: its first half adds zeros.
     B3_MFMA(Ah, 0, Bh, 0);
     B3_MFMA(Ah, 0, Bh, 1);
     B3_MFMA(Ah, 0, Bh, 2);
     B3_MFMA(Ah, 0, Bh, 3);
     B3_MFMA(Ah, 1, Bh, 0);
     B3_MFMA(Ah, 1, Bh, 1);
     B3_MFMA(Ah, 1, Bh, 2);
     B3_MFMA(Ah, 1, Bh, 3);
     B3_MFMA(Ah, 2, Bh, 0);
     B3_MFMA(Ah, 2, Bh, 1);
     B3_MFMA(Ah, 2, Bh, 2);
     B3_MFMA(Ah, 2, Bh, 3);
     B3_MFMA(Ah, 3, Bh, 0);
     B3_MFMA(Ah, 3, Bh, 1);
     B3_MFMA(Ah, 3, Bh, 2);
     B3_MFMA(Ah, 3, Bh, 3);
     B3_MFMA(Ah, 0, Bl, 0);
     B3_MFMA(Ah, 0, Bl, 1);
     B3_MFMA(Ah, 0, Bl, 2);
     B3_MFMA(Ah, 0, Bl, 3);
     B3_MFMA(Ah, 1, Bl, 0);
     B3_MFMA(Ah, 1, Bl, 1);
     B3_MFMA(Ah, 1, Bl, 2);
     B3_MFMA(Ah, 1, Bl, 3);
     B3_MFMA(Ah, 2, Bl, 0);
     B3_MFMA(Ah, 2, Bl, 1);
     B3_MFMA(Ah, 2, Bl, 2);
     B3_MFMA(Ah, 2, Bl, 3);
     B3_MFMA(Ah, 3, Bl, 0);
     B3_MFMA(Ah, 3, Bl, 1);
     B3_MFMA(Ah, 3, Bl, 2);
     B3_MFMA(Ah, 3, Bl, 3);
     B3_MFMA(Al, 0, Bh, 0);
     B3_MFMA(Al, 0, Bh, 1);
     B3_MFMA(Al, 0, Bh, 2);
     B3_MFMA(Al, 0, Bh, 3);
     B3_MFMA(Al, 1, Bh, 0);
     B3_MFMA(Al, 1, Bh, 1);
     B3_MFMA(Al, 1, Bh, 2);
     B3_MFMA(Al, 1, Bh, 3);
     B3_MFMA(Al, 2, Bh, 0);
     B3_MFMA(Al, 2, Bh, 1);
     B3_MFMA(Al, 2, Bh, 2);
     B3_MFMA(Al, 2, Bh, 3);
     B3_MFMA(Al, 3, Bh, 0);
     B3_MFMA(Al, 3, Bh, 1);
     B3_MFMA(Al, 3, Bh, 2);
     B3_MFMA(Al, 3, Bh, 3);
    // Three slots, LDS-DMA two K-steps deep: iteration k reads the fragments of step k+1 from slot (k+1) % 3 while step k+2 is
    // still landing in slot (k+2) % 3 and step k+3 is issued into slot k % 3 (step k lives in registers since iteration k-1).
    int nxt = 1, fill = 0;
    for (int k = 0; k < nks; ++k) {
      // all but my newest 12 LDS-DMA pieces (= step k+2's, issued an iteration ago) have landed: step k+1 is there, issued TWO
      // iterations ago (~5 us: with one iteration of distance the wait stalled on MALL / HBM misses, -10 %); every fragment read
      // has returned (a real s_waitcnt the compiler can see, so that it does not wait for "unknown" loop-carried reads behind this
      // iteration's first ones); the barrier: ... everybody's; every wave has consumed the fragments of step k
      B3X_SYNC();
      asm volatile("" ::: "memory");
      const char* aa_ = smem + nxt * SLAB + ra0;
      const char* bb_ = smem + nxt * SLAB + rb0;
      B3X_READ(Al[0], aa_, 2 * PLANE_A + 0 * 1024); B3_MFMA(Ah, 0, Bm, 0);
      B3X_READ(Al[1], aa_, 2 * PLANE_A + 1 * 1024); B3_MFMA(Ah, 0, Bm, 1);
      B3X_READ(Al[2], aa_, 2 * PLANE_A + 2 * 1024); B3_MFMA(Ah, 0, Bm, 2);
      B3X_READ(Al[3], aa_, 2 * PLANE_A + 3 * 1024); B3_MFMA(Ah, 0, Bm, 3);
      B3X_READ(Bl[0], bb_, 2 * PLANE_B + 0 * 1024); B3_MFMA(Ah, 1, Bm, 0);
      B3X_READ(Bl[1], bb_, 2 * PLANE_B + 1 * 1024); B3_MFMA(Ah, 1, Bm, 1);
      B3X_READ(Bl[2], bb_, 2 * PLANE_B + 2 * 1024); B3_MFMA(Ah, 1, Bm, 2);
      B3X_READ(Bl[3], bb_, 2 * PLANE_B + 3 * 1024); B3_MFMA(Ah, 1, Bm, 3);
      B3X_PIECE(fill, 0); B3_MFMA(Ah, 2, Bm, 0);
      B3X_PIECE(fill, 1); B3_MFMA(Ah, 2, Bm, 1);
      B3X_PIECE(fill, 2); B3_MFMA(Ah, 2, Bm, 2);
      B3X_PIECE(fill, 3); B3_MFMA(Ah, 2, Bm, 3);
      B3X_PIECE(fill, 4); B3_MFMA(Ah, 3, Bm, 0);
      B3X_PIECE(fill, 5); B3_MFMA(Ah, 3, Bm, 1);
      B3X_PIECE(fill, 6); B3_MFMA(Ah, 3, Bm, 2);
      B3X_PIECE(fill, 7); B3_MFMA(Ah, 3, Bm, 3);
      B3X_READ(Ah[0], aa_, 0 * PLANE_A + 0 * 1024); B3_MFMA(Am, 0, Bh, 0);
      B3X_READ(Ah[1], aa_, 0 * PLANE_A + 1 * 1024); B3_MFMA(Am, 0, Bh, 1);
      B3X_READ(Ah[2], aa_, 0 * PLANE_A + 2 * 1024); B3_MFMA(Am, 0, Bh, 2);
      B3X_READ(Ah[3], aa_, 0 * PLANE_A + 3 * 1024); B3_MFMA(Am, 0, Bh, 3);
      B3X_PIECE(fill, 8); B3_MFMA(Am, 1, Bh, 0);
      B3X_PIECE(fill, 9); B3_MFMA(Am, 1, Bh, 1);
      B3X_PIECE(fill, 10); B3_MFMA(Am, 1, Bh, 2);
      B3X_PIECE(fill, 11); B3_MFMA(Am, 1, Bh, 3);
       B3_MFMA(Am, 2, Bh, 0);
       B3_MFMA(Am, 2, Bh, 1);
       B3_MFMA(Am, 2, Bh, 2);
       B3_MFMA(Am, 2, Bh, 3);
       B3_MFMA(Am, 3, Bh, 0);
       B3_MFMA(Am, 3, Bh, 1);
       B3_MFMA(Am, 3, Bh, 2);
       B3_MFMA(Am, 3, Bh, 3);
      B3X_READ(Bh[0], bb_, 0 * PLANE_B + 0 * 1024); B3_MFMA(Am, 0, Bm, 0);
      B3X_READ(Bh[1], bb_, 0 * PLANE_B + 1 * 1024); B3_MFMA(Am, 0, Bm, 1);
      B3X_READ(Bh[2], bb_, 0 * PLANE_B + 2 * 1024); B3_MFMA(Am, 0, Bm, 2);
      B3X_READ(Bh[3], bb_, 0 * PLANE_B + 3 * 1024); B3_MFMA(Am, 0, Bm, 3);
       B3_MFMA(Am, 1, Bm, 0);
       B3_MFMA(Am, 1, Bm, 1);
       B3_MFMA(Am, 1, Bm, 2);
       B3_MFMA(Am, 1, Bm, 3);
       B3_MFMA(Am, 2, Bm, 0);
       B3_MFMA(Am, 2, Bm, 1);
       B3_MFMA(Am, 2, Bm, 2);
       B3_MFMA(Am, 2, Bm, 3);
       B3_MFMA(Am, 3, Bm, 0);
       B3_MFMA(Am, 3, Bm, 1);
       B3_MFMA(Am, 3, Bm, 2);
       B3_MFMA(Am, 3, Bm, 3);
       B3_MFMA(Ah, 0, Bh, 0);
      B3X_READ(Am[0], aa_, 1 * PLANE_A + 0 * 1024); B3_MFMA(Ah, 0, Bh, 1);
      B3X_READ(Am[1], aa_, 1 * PLANE_A + 1 * 1024); B3_MFMA(Ah, 0, Bh, 2);
      B3X_READ(Am[2], aa_, 1 * PLANE_A + 2 * 1024); B3_MFMA(Ah, 0, Bh, 3);
      B3X_READ(Am[3], aa_, 1 * PLANE_A + 3 * 1024); B3_MFMA(Ah, 1, Bh, 0);
      B3X_READ(Bm[0], bb_, 1 * PLANE_B + 0 * 1024); B3_MFMA(Ah, 1, Bh, 1);
      B3X_READ(Bm[1], bb_, 1 * PLANE_B + 1 * 1024); B3_MFMA(Ah, 1, Bh, 2);
      B3X_READ(Bm[2], bb_, 1 * PLANE_B + 2 * 1024); B3_MFMA(Ah, 1, Bh, 3);
      B3X_READ(Bm[3], bb_, 1 * PLANE_B + 3 * 1024); B3_MFMA(Ah, 2, Bh, 0);
       B3_MFMA(Ah, 2, Bh, 1);
      B3X_PREPA(0); B3_MFMA(Ah, 2, Bh, 2);
       B3_MFMA(Ah, 2, Bh, 3);
       B3_MFMA(Ah, 3, Bh, 0);
      B3X_PREPA(1); B3_MFMA(Ah, 3, Bh, 1);
       B3_MFMA(Ah, 3, Bh, 2);
       B3_MFMA(Ah, 3, Bh, 3);
       B3_MFMA(Ah, 0, Bl, 0);
      B3X_PREPB(0); B3_MFMA(Ah, 0, Bl, 1);
       B3_MFMA(Ah, 0, Bl, 2);
       B3_MFMA(Ah, 0, Bl, 3);
       B3_MFMA(Ah, 1, Bl, 0);
      B3X_PREPB(1); B3_MFMA(Ah, 1, Bl, 1);
       B3_MFMA(Ah, 1, Bl, 2);
       B3_MFMA(Ah, 1, Bl, 3);
       B3_MFMA(Ah, 2, Bl, 0);
      B3X_ADVANCE(); B3_MFMA(Ah, 2, Bl, 1);
       B3_MFMA(Ah, 2, Bl, 2);
       B3_MFMA(Ah, 2, Bl, 3);
       B3_MFMA(Ah, 3, Bl, 0);
       B3_MFMA(Ah, 3, Bl, 1);
       B3_MFMA(Ah, 3, Bl, 2);
       B3_MFMA(Ah, 3, Bl, 3);
       B3_MFMA(Al, 0, Bh, 0);
       B3_MFMA(Al, 0, Bh, 1);
       B3_MFMA(Al, 0, Bh, 2);
       B3_MFMA(Al, 0, Bh, 3);
       B3_MFMA(Al, 1, Bh, 0);
       B3_MFMA(Al, 1, Bh, 1);
       B3_MFMA(Al, 1, Bh, 2);
       B3_MFMA(Al, 1, Bh, 3);
       B3_MFMA(Al, 2, Bh, 0);
       B3_MFMA(Al, 2, Bh, 1);
       B3_MFMA(Al, 2, Bh, 2);
       B3_MFMA(Al, 2, Bh, 3);
       B3_MFMA(Al, 3, Bh, 0);
       B3_MFMA(Al, 3, Bh, 1);
       B3_MFMA(Al, 3, Bh, 2);
       B3_MFMA(Al, 3, Bh, 3);
      nxt = nxt == 2 ? 0 : nxt + 1;
      fill = fill == 2 ? 0 : fill + 1;
    }
  }
#undef B3_DS_READ
#undef B3_READ_PLANE
#undef B3_MFMA

  if (p.bn_part != nullptr) {   // BatchNorm batch statistics of this tile (common.h: cs_tile_bn_partials)
    __syncthreads();
    int colv[TN];
    float bvv[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      colv[u] = wn * 128 + u * 32 + l31;
      bvv[u] = (p.bias != nullptr && n0 + colv[u] < p.N) ? p.bias[n0 + colv[u]] : 0.f;
    }
    const int rbase = m0 + wm * 128 + 4 * h;
    cs_tile_bn_partials<TN, TM * 16, 2, false>(
        (float*)smem, BN, colv, h == 0, wm, min(BM, p.M - m0),
        [&](int j, int i) { return acc[i >> 4][j][i & 15] + bvv[j]; },
        [&](int i) { return rbase + (i >> 4) * 32 + (i & 3) + 8 * ((i & 15) >> 2) < p.M; }, p.bn_part + (long long)tile_m * 3 * p.N, p.N, n0);
  }
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int col = n0 + wn * 128 + u * 32 + l31;
      const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < p.M) {
          float* dst = p.C + (long long)row * p.ldc + col;
          if (col < p.N) {
            float v = acc[t][u][r] + bv;
            if (p.accumulate) v += *dst;
            if (p.residual != nullptr) v += p.residual[(long long)row * p.ldr + col];
            if (p.relu) v = fmaxf(v, 0.f);
            *dst = v;
          } else if (col < p.zero_to) {
            *dst = 0.f;
          }
        }
      }
    }
}


// ---- backward-weight: dW[o][tap][c] = sum_p dy[p][o] * x[pix(p, tap)][c] as a "TN" GEMM, M = Cout, N = taps * Cin (all filter
// taps side by side), reduction over the output pixels p (split over blockIdx.z, partial slabs reduced afterwards).
// Both operands are pixel-major in memory (the reduction index is the slow one), so the LDS images are [16 pixels][256 columns]
// per plane and the MFMA fragments (8 consecutive k for one column) are read with the hardware transpose ds_read_b64_tr_b16:
// per 16-lane group a block of 4 pixels x 16 columns, lane 4q + p supplying the address of pixel-row q, columns 4p .. 4p+3.
// The 16-byte chunks of a pixel row are stored at chunk ^ ((pixel & 3) << 2): the four pixel rows of a block then hit four
// different 64-byte bank groups (conflict-free; the swizzle is applied to the LDS-DMA SOURCE chunk, the destination is lane-linear).
// Structure otherwise as igemm_b3w_kernel: 4 waves, one per SIMD, 256 x 256 tile, register-pipelined, pinned slot schedule.
struct B3TArgs {
  const u16* dy;  long long dy_plane; int ldo;     // [P][ldo] planes, ldo = roundup(Cout, 8)
  const u16* x;   long long x_plane;  int ldx;     // [B*H*W][ldx] planes
  float* C;       int ldc; long long c_split_stride;
  int M, N, Cin, taps;                             // M = Cout, N = taps * Cin
  int P, rows_per_split;                           // output pixels; multiple of 16 per split
  int H, W, Ho, Wo, kw, stride, pad, dil;
  int step_b, step_qy, step_rx;                    // 16 pixels = step_b images + step_qy rows + step_rx pixels
  int tilesM, tilesN;
  const float* zero;
};

// (differential timing builds of the backward-weight kernel: -DB3T_NO_PREP / _NO_DMA / _NO_SYNC drop one ingredient of its K loop --
//  wrong results, the time difference is that ingredient's cost)
#ifdef B3T_NO_DMA
#define B3T_PIECE(f, i) do {} while (0)
#else
#define B3T_PIECE(f, i) piece(f, i)
#endif
#ifdef B3T_NO_PREP
#define B3T_PREPA() do {} while (0)
#define B3T_PREPB(j) do {} while (0)
#else
#define B3T_PREPA() prepA()
#define B3T_PREPB(j) prepB(j)
#endif
#ifdef B3T_NO_SYNC
#define B3T_SYNC() do {} while (0)
#else
#define B3T_SYNC()                                           \
  do {                                                       \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         \
    __builtin_amdgcn_s_barrier();                            \
  } while (0)
#endif
__global__ __launch_bounds__(256, 1) void igemm_b3t_kernel(const B3TArgs p) {
  constexpr int TM = 4, TN = 4;
  constexpr int PLANE = 16 * 256 * 2;             // bytes of one operand plane image: 16 pixel rows x 256 columns of bf16
  constexpr int SLAB = 6 * PLANE;                 // A planes 0..2, then B planes 0..2
  __shared__ __attribute__((aligned(16))) char smem[3 * SLAB];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  // one-dimensional grid over (pixel slab, tile): the blocks that share an XCD (equal blockIdx.x % 8) get a contiguous run of it, so
  // that all 2 x 26 tiles of a slab run on ONE XCD and re-read the slab's dy / x rows from that XCD's L2 (with the slab index in
  // blockIdx.z the tiles of a slab were dealt round-robin over the eight XCDs: every L2 fetched every slab, 8x the operand bytes)
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int ntile = p.tilesM * p.tilesN;
  const int split = swz / ntile, tix = swz - split * ntile;
  const int tile_m = tix / p.tilesN, tile_n = tix - tile_m * p.tilesN;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int r_begin = split * p.rows_per_split;
  const int r_end = min(r_begin + p.rows_per_split, p.P);
  const int nks = (max(r_end - r_begin, 0) + 15) >> 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- staging: chunk q = j * 256 + tid of a plane image: pixel row krow = q >> 5, position q & 31 (lane-linear LDS-DMA),
  //      logical 8-column chunk = position ^ ((krow & 3) << 2)
  const u16* zero = (const u16*)p.zero;
  int arow[2];            // pixel row of the NEXT un-prepared K-step
  int acol[2];            // column offset into the dy row, or -1 if outside [0, ldo)
  int brow[2], tb[2], ty[2], tx[2], bky[2], bkx[2], bch[2];
  bool bcv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = j * 256 + tid, krow = q >> 5, cl = (q & 31) ^ ((krow & 3) << 2);
    arow[j] = r_begin + krow;
    acol[j] = (m0 + cl * 8) < p.ldo ? m0 + cl * 8 : -1;
    brow[j] = r_begin + krow;
    const int n = n0 + cl * 8;
    bcv[j] = n < p.N;
    const int tap = bcv[j] ? n / p.Cin : 0;
    bch[j] = n - tap * p.Cin;
    bky[j] = tap / p.kw;
    bkx[j] = tap - bky[j] * p.kw;
    const int r = brow[j] < p.P ? brow[j] : 0;
    const int hw = p.Ho * p.Wo;
    tb[j] = r / hw;
    const int rem = r - tb[j] * hw;
    ty[j] = rem / p.Wo;
    tx[j] = rem - ty[j] * p.Wo;
  }
  const u16* PA[2][3];
  const u16* PB[2][3];
  const long long apl = p.dy_plane, xpl = p.x_plane;
  auto prepA = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = (int)(arow[j] < r_end) & (int)(acol[j] >= 0);
      const u16* q = p.dy + ((long long)(ok ? arow[j] : 0) * p.ldo + (ok ? acol[j] : 0));
      const long long st = ok ? apl : 0;
      PA[j][0] = ok ? q : zero;
      PA[j][1] = PA[j][0] + st;
      PA[j][2] = PA[j][1] + st;
      arow[j] += 16;
    }
  };
  auto prepB = [&](const int j) {
    const int iy = ty[j] * p.stride - p.pad + bky[j] * p.dil;
    const int ix = tx[j] * p.stride - p.pad + bkx[j] * p.dil;
    const bool ok = (int)(brow[j] < r_end) & (int)bcv[j] & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);
    const u16* q = p.x + (unsigned)(ok ? ((tb[j] * p.H + iy) * p.W + ix) * p.ldx + bch[j] : 0);
    const long long st = ok ? xpl : 0;
    PB[j][0] = ok ? q : zero;
    PB[j][1] = PB[j][0] + st;
    PB[j][2] = PB[j][1] + st;
    // advance this row by 16 pixels
    brow[j] += 16;
    tx[j] += p.step_rx;
    ty[j] += p.step_qy;
    const bool cx = tx[j] >= p.Wo;
    tx[j] -= cx ? p.Wo : 0;
    ty[j] += cx ? 1 : 0;
    const bool cy = ty[j] >= p.Ho;
    ty[j] -= cy ? p.Ho : 0;
    tb[j] += p.step_b + (cy ? 1 : 0);
  };
  auto prep = [&]() { prepA(); prepB(0); prepB(1); };
  auto piece = [&](const int buf, const int i) {     // i = 4 * plane + {A0, A1, B0, B1}
    char* s = smem + buf * SLAB;
    const int pl = i >> 2, w = i & 3;
    if (w < 2) glds16(PA[w][pl], s + pl * PLANE + (w * 256 + wave * 64) * 16);
    else glds16(PB[w - 2][pl], s + (3 + pl) * PLANE + ((w - 2) * 256 + wave * 64) * 16);
  };
  auto issue = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < 12; ++i) piece(buf, i);
  };

  // ---- fragment addresses (transposed reads): lane = 16 g + 4 q + pp inside its 32-lane half
  int ra[TM], rb[TN];
  {
    const int g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    const int rowb = (8 * h + q) * 512;               // pixel row 8h + q (+ 4 for the second read: + 2048 bytes)
#pragma unroll
    for (int t = 0; t < TM; ++t) ra[t] = rowb + ((wm * 16 + ((t ^ q) << 2) + 2 * g1 + (pp >> 1)) << 4) + ((pp & 1) << 3);
#pragma unroll
    for (int u = 0; u < TN; ++u) rb[u] = 3 * PLANE + rowb + ((wn * 16 + ((u ^ q) << 2) + 2 * g1 + (pp >> 1)) << 4) + ((pp & 1) << 3);
  }
  bf16x8 Ah[TM], Bh[TN], Al[TM], Bl[TN], Am[TM], Bm[TN];
#define T_READ(dst, base, off)                                                                              \
  do {                                                                                                      \
    const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((base) + (off)));                \
    const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((base) + (off) + 2048));         \
    typedef short s16x8_ __attribute__((ext_vector_type(8)));                                               \
    const s16x8_ v_ = {lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]};                       \
    dst = __builtin_bit_cast(bf16x8, v_);                                                                   \
  } while (0)
#define B3_MFMA(A_, t_, B_, u_)                                                                                \
  do {                                                                                                        \
    acc[t_][u_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[t_], B_[u_], acc[t_][u_], 0, 0, 0);               \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)

  if (nks > 0) {
    prep();
    issue(0);
    prep();
    if (nks > 1) issue(1);
    prep();
    if (nks > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      const char* aa_[TM]; const char* bb_[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) { aa_[t] = smem + ra[t]; bb_[t] = smem + rb[t]; }
#pragma unroll
      for (int t = 0; t < TM; ++t) { T_READ(Al[t], aa_[t], 2 * PLANE); T_READ(Bl[t], bb_[t], 2 * PLANE); }
#pragma unroll
      for (int t = 0; t < TM; ++t) { T_READ(Ah[t], aa_[t], 0 * PLANE); T_READ(Bh[t], bb_[t], 0 * PLANE); }
#pragma unroll
      for (int t = 0; t < TM; ++t) { T_READ(Am[t], aa_[t], 1 * PLANE); T_READ(Bm[t], bb_[t], 1 * PLANE); }
    }
    int nxt = 1, fill = 2;
    for (int k = 0; k < nks; ++k) {
      B3T_SYNC();   // my LDS-DMA of K-step k+1 has landed (issued a K-step ago) + barrier
      asm volatile("" ::: "memory");
      const char* aa_[TM]; const char* bb_[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) { aa_[t] = smem + nxt * SLAB + ra[t]; bb_[t] = smem + nxt * SLAB + rb[t]; }
      B3T_PIECE(fill, 0); B3_MFMA(Ah, 0, Bh, 0);
       B3_MFMA(Ah, 0, Bh, 1);
       B3_MFMA(Ah, 0, Bh, 2);
      B3T_PIECE(fill, 1); B3_MFMA(Ah, 0, Bh, 3);
       B3_MFMA(Ah, 1, Bh, 0);
       B3_MFMA(Ah, 1, Bh, 1);
      B3T_PIECE(fill, 2); B3_MFMA(Ah, 1, Bh, 2);
       B3_MFMA(Ah, 1, Bh, 3);
       B3_MFMA(Ah, 2, Bh, 0);
      B3T_PIECE(fill, 3); B3_MFMA(Ah, 2, Bh, 1);
       B3_MFMA(Ah, 2, Bh, 2);
       B3_MFMA(Ah, 2, Bh, 3);
      B3T_PIECE(fill, 4); B3_MFMA(Ah, 3, Bh, 0);
       B3_MFMA(Ah, 3, Bh, 1);
       B3_MFMA(Ah, 3, Bh, 2);
      B3T_PIECE(fill, 5); B3_MFMA(Ah, 3, Bh, 3);
       B3_MFMA(Ah, 0, Bl, 0);
       B3_MFMA(Ah, 0, Bl, 1);
      B3T_PIECE(fill, 6); B3_MFMA(Ah, 0, Bl, 2);
       B3_MFMA(Ah, 0, Bl, 3);
       B3_MFMA(Ah, 1, Bl, 0);
      B3T_PIECE(fill, 7); B3_MFMA(Ah, 1, Bl, 1);
       B3_MFMA(Ah, 1, Bl, 2);
       B3_MFMA(Ah, 1, Bl, 3);
      B3T_PIECE(fill, 8); B3_MFMA(Ah, 2, Bl, 0);
       B3_MFMA(Ah, 2, Bl, 1);
       B3_MFMA(Ah, 2, Bl, 2);
      B3T_PIECE(fill, 9); B3_MFMA(Ah, 2, Bl, 3);
       B3_MFMA(Ah, 3, Bl, 0);
       B3_MFMA(Ah, 3, Bl, 1);
      B3T_PIECE(fill, 10); B3_MFMA(Ah, 3, Bl, 2);
       B3_MFMA(Ah, 3, Bl, 3);
       B3_MFMA(Al, 0, Bh, 0);
      B3T_PIECE(fill, 11); B3_MFMA(Al, 0, Bh, 1);
       B3_MFMA(Al, 0, Bh, 2);
       B3_MFMA(Al, 0, Bh, 3);
      B3T_PREPA(); B3_MFMA(Al, 1, Bh, 0);
       B3_MFMA(Al, 1, Bh, 1);
      B3T_PREPB(0); B3_MFMA(Al, 1, Bh, 2);
       B3_MFMA(Al, 1, Bh, 3);
      B3T_PREPB(1); B3_MFMA(Al, 2, Bh, 0);
       B3_MFMA(Al, 2, Bh, 1);
       B3_MFMA(Al, 2, Bh, 2);
       B3_MFMA(Al, 2, Bh, 3);
       B3_MFMA(Al, 3, Bh, 0);
       B3_MFMA(Al, 3, Bh, 1);
       B3_MFMA(Al, 3, Bh, 2);
       B3_MFMA(Al, 3, Bh, 3);
      T_READ(Al[0], aa_[0], 2 * PLANE); B3_MFMA(Ah, 0, Bm, 0);
      T_READ(Al[1], aa_[1], 2 * PLANE); B3_MFMA(Ah, 0, Bm, 1);
      T_READ(Al[2], aa_[2], 2 * PLANE); B3_MFMA(Ah, 0, Bm, 2);
      T_READ(Al[3], aa_[3], 2 * PLANE); B3_MFMA(Ah, 0, Bm, 3);
      T_READ(Bl[0], bb_[0], 2 * PLANE); B3_MFMA(Ah, 1, Bm, 0);
      T_READ(Bl[1], bb_[1], 2 * PLANE); B3_MFMA(Ah, 1, Bm, 1);
      T_READ(Bl[2], bb_[2], 2 * PLANE); B3_MFMA(Ah, 1, Bm, 2);
      T_READ(Bl[3], bb_[3], 2 * PLANE); B3_MFMA(Ah, 1, Bm, 3);
       B3_MFMA(Ah, 2, Bm, 0);
       B3_MFMA(Ah, 2, Bm, 1);
       B3_MFMA(Ah, 2, Bm, 2);
       B3_MFMA(Ah, 2, Bm, 3);
       B3_MFMA(Ah, 3, Bm, 0);
       B3_MFMA(Ah, 3, Bm, 1);
       B3_MFMA(Ah, 3, Bm, 2);
       B3_MFMA(Ah, 3, Bm, 3);
      T_READ(Ah[0], aa_[0], 0 * PLANE); B3_MFMA(Am, 0, Bh, 0);
      T_READ(Ah[1], aa_[1], 0 * PLANE); B3_MFMA(Am, 0, Bh, 1);
      T_READ(Ah[2], aa_[2], 0 * PLANE); B3_MFMA(Am, 0, Bh, 2);
      T_READ(Ah[3], aa_[3], 0 * PLANE); B3_MFMA(Am, 0, Bh, 3);
       B3_MFMA(Am, 1, Bh, 0);
       B3_MFMA(Am, 1, Bh, 1);
       B3_MFMA(Am, 1, Bh, 2);
       B3_MFMA(Am, 1, Bh, 3);
       B3_MFMA(Am, 2, Bh, 0);
       B3_MFMA(Am, 2, Bh, 1);
       B3_MFMA(Am, 2, Bh, 2);
       B3_MFMA(Am, 2, Bh, 3);
       B3_MFMA(Am, 3, Bh, 0);
       B3_MFMA(Am, 3, Bh, 1);
       B3_MFMA(Am, 3, Bh, 2);
       B3_MFMA(Am, 3, Bh, 3);
      T_READ(Bh[0], bb_[0], 0 * PLANE); B3_MFMA(Am, 0, Bm, 0);
      T_READ(Bh[1], bb_[1], 0 * PLANE); B3_MFMA(Am, 0, Bm, 1);
      T_READ(Bh[2], bb_[2], 0 * PLANE); B3_MFMA(Am, 0, Bm, 2);
      T_READ(Bh[3], bb_[3], 0 * PLANE); B3_MFMA(Am, 0, Bm, 3);
      T_READ(Am[0], aa_[0], 1 * PLANE); B3_MFMA(Am, 1, Bm, 0);
       B3_MFMA(Am, 1, Bm, 1);
       B3_MFMA(Am, 1, Bm, 2);
       B3_MFMA(Am, 1, Bm, 3);
      T_READ(Am[1], aa_[1], 1 * PLANE); B3_MFMA(Am, 2, Bm, 0);
       B3_MFMA(Am, 2, Bm, 1);
       B3_MFMA(Am, 2, Bm, 2);
       B3_MFMA(Am, 2, Bm, 3);
      T_READ(Am[2], aa_[2], 1 * PLANE); B3_MFMA(Am, 3, Bm, 0);
       B3_MFMA(Am, 3, Bm, 1);
       B3_MFMA(Am, 3, Bm, 2);
       B3_MFMA(Am, 3, Bm, 3);
      T_READ(Am[3], aa_[3], 1 * PLANE);
      T_READ(Bm[0], bb_[0], 1 * PLANE);
      T_READ(Bm[1], bb_[1], 1 * PLANE);
      T_READ(Bm[2], bb_[2], 1 * PLANE);
      T_READ(Bm[3], bb_[3], 1 * PLANE);
      nxt = nxt == 2 ? 0 : nxt + 1;
      fill = fill == 2 ? 0 : fill + 1;
    }
  }
#undef T_READ
#undef B3_MFMA

  float* cout = p.C + (long long)split * p.c_split_stride;
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int col = n0 + wn * 128 + u * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < p.M && col < p.N) cout[(long long)row * p.ldc + col] = acc[t][u][r];
      }
    }
}

// out[i] = sum_s slab[s][i]  (fixed order: deterministic)
__global__ void b3_reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n4, int splits, long long stride4) {
  const f32x4* s = (const f32x4*)slabs;
  f32x4* o = (f32x4*)out;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 a = s[i];
    for (int k = 1; k < splits; ++k) a += s[i + k * stride4];
    o[i] = a;
  }
}

const float* zero_page_b3() {
  static const float* z[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!z[dev]) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_page_b3)) == hipSuccess) z[dev] = (const float*)q;
  }
  return z[dev];
}

int g_b3_tile = 0;   // tuning hook: 0 = heuristic, 1 = 256x256 (8 waves), 2 / 7 = 256x128, 3 = 128x256, 4 = 256x192, 5 = 256x96, 6 = 256x64, 9 = 256x256 (4 waves, register-pipelined)

template <int TM, int TN, int WGM, int WGN, int NBUF = 3>
void launch_b3(B3Args a, hipStream_t st) {
  constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
  a.tilesM = (a.M + BM - 1) / BM;
  const int ncols = a.zero_to > a.N ? a.zero_to : a.N;
  a.tilesN = (ncols + BN - 1) / BN;
  a.zero = zero_page_b3();
  hipLaunchKernelGGL((igemm_b3_kernel<TM, TN, WGM, WGN, NBUF>), dim3(a.tilesM * a.tilesN), dim3(512), 0, st, a);
}

int pick_b3_tile(int N) {
  if (g_b3_tile) return g_b3_tile;
  if (N <= 64) return 6;
  if (N <= 96) return 5;
  if (N <= 128) return 2;
  if (N <= 192) return 4;
  if (N == 192) return 4;
  return 9;   // 256 x 256, one wave per SIMD, register-pipelined
}

int run_b3(const B3Args& a, hipStream_t st) {
  int tile = pick_b3_tile(a.zero_to > a.N ? a.zero_to : a.N);
  // the register-pipelined kernel addresses each operand through one 32-bit-offset buffer resource over its three planes
  if (tile == 9 && (a.a_plane * 6 >= (1ll << 32) - 64 || a.w_plane * 6 >= (1ll << 32) - 64)) tile = 1;
  if (a.blocked) {
    CS_REQUIRE(a.a_plane * 6 < (1ll << 32) - 64 && a.w_plane * 6 < (1ll << 32) - 64, "bf16x3 blocked planes: an operand's three planes must stay below 4 GB");
    tile = 9;
  }
  switch (tile) {
    case 1: launch_b3<4, 2, 2, 4>(a, st); break;   // 256 x 256, wave tile 128 x 64
    case 2: launch_b3<2, 2, 4, 2, 2>(a, st); break;   // 256 x 128, wave tile 64 x 64 (two slots: two blocks per CU)
    case 7: launch_b3<2, 2, 4, 2, 3>(a, st); break;   // 256 x 128, three slots
    case 8: launch_b3<4, 2, 2, 4, 2>(a, st); break;   // 256 x 256, two slots (A/B measurement)
    case 3: launch_b3<2, 2, 2, 4>(a, st); break;   // 128 x 256
    case 4: launch_b3<2, 3, 4, 2>(a, st); break;   // 256 x 192, wave tile 64 x 96
    case 5: launch_b3<1, 3, 8, 1, 2>(a, st); break;   // 256 x 96,  wave tile 32 x 96
    case 6: launch_b3<1, 2, 8, 1, 2>(a, st); break;   // 256 x 64,  wave tile 32 x 64
    case 9: {                                      // 256 x 256, one wave per SIMD, register-pipelined
      B3Args q = a;
      q.tilesM = (q.M + 255) / 256;
      q.tilesN = ((q.zero_to > q.N ? q.zero_to : q.N) + 255) / 256;
      q.zero = zero_page_b3();
      if (q.blocked) hipLaunchKernelGGL(igemm_b3w_kernel<true>, dim3(q.tilesM * q.tilesN), dim3(256), 0, st, q);
      else hipLaunchKernelGGL(igemm_b3w_kernel<false>, dim3(q.tilesM * q.tilesN), dim3(256), 0, st, q);
      break;
    }
    default: catseg_set_error("bf16x3: unknown tile"); return CATSEG_EINVAL;
  }
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

}  // namespace

extern "C" int catseg_debug_set_b3_tile(int t) {
  g_b3_tile = t;
  return CATSEG_OK;
}

extern "C" size_t catseg_split3_elems(long long rows, int C) { return (size_t)rows * ((C + 7) & ~7); }

// planes[p][row][ldp] (bf16, ldp = roundup(C, 8)); `planes` must hold 3 * catseg_split3_elems(rows, C) 16-bit elements
extern "C" int catseg_split3(const float* x, int ld, long long rows, int C, void* planes, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && ld >= C && cs_aligned16(planes), "split3: bad args");
  const int ldp = (C + 7) & ~7;
  const long long n = rows * (ldp >> 3);
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ld, rows, C, ldp, (u16*)planes, rows * ldp);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// OHWI weights -> planes of [Cin][taps][ldp = roundup(O, 8)]
extern "C" int catseg_split3_weight_t(const float* w, int O, int taps, int Cin, void* planes, catseg_stream_t stream) {
  CS_REQUIRE(O > 0 && taps > 0 && Cin > 0 && cs_aligned16(planes), "split3_weight_t: bad args");
  const int ldp = (O + 7) & ~7;
  const long long n = (long long)Cin * taps * ldp;
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split3_wt_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, O, taps, Cin, ldp, (u16*)planes, n);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// y = conv(x, w) (+ bias) from pre-split planes: x_planes = catseg_split3 of x (rows = B*H*W, C = Cin),
// w_planes = catseg_split3 of the OHWI weights viewed as [Cout][kh*kw*Cin]
extern "C" int catseg_conv2d_fwd_bf16x3(const catseg_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                                        float* y, int zero_to, catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->Cin % 8 == 0 && d->kh * d->kw <= 32, "conv fwd bf16x3: needs Cin % 8 == 0, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(w_planes) && cs_aligned16(y) && zero_to <= d->ldy, "conv fwd bf16x3: alignment");
  CS_REQUIRE((long long)d->B * d->H * d->W * d->Cin < (1ll << 31) && (long long)d->Cout * d->kh * d->kw * d->Cin < (1ll << 31), "conv fwd bf16x3: 32-bit offsets");
  B3Args a = {};
  a.a = (const u16*)x_planes; a.lda = d->Cin; a.a_plane = (long long)d->B * d->H * d->W * d->Cin;
  a.w = (const u16*)w_planes; a.ldw = d->kh * d->kw * d->Cin; a.w_plane = (long long)d->Cout * a.ldw;
  a.C = y; a.ldc = d->ldy; a.bias = bias;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Cin = d->Cin; a.taps = d->kh * d->kw;
  a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.sign = 1; a.zero_to = zero_to;
  return run_b3(a, (hipStream_t)stream);
}

// as catseg_conv2d_fwd_bf16x3, plus per-(M-tile, channel) BatchNorm partials from the epilogue (see catseg_conv2d_fwd_bnstats)
extern "C" int catseg_conv2d_fwd_bf16x3_bnstats(const catseg_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                                                float* y, int zero_to, float* bn_part, size_t bn_part_floats, int* tile_rows, int* n_tiles,
                                                catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->Cin % 8 == 0 && d->kh * d->kw <= 32, "conv fwd bf16x3: needs Cin % 8 == 0, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(w_planes) && cs_aligned16(y) && zero_to <= d->ldy && tile_rows && n_tiles, "conv fwd bf16x3: bad args");
  CS_REQUIRE((long long)d->B * d->H * d->W * d->Cin < (1ll << 31) && (long long)d->Cout * d->kh * d->kw * d->Cin < (1ll << 31), "conv fwd bf16x3: 32-bit offsets");
  B3Args a = {};
  a.a = (const u16*)x_planes; a.lda = d->Cin; a.a_plane = (long long)d->B * d->H * d->W * d->Cin;
  a.w = (const u16*)w_planes; a.ldw = d->kh * d->kw * d->Cin; a.w_plane = (long long)d->Cout * a.ldw;
  a.C = y; a.ldc = d->ldy; a.bias = bias;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Cin = d->Cin; a.taps = d->kh * d->kw;
  a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.sign = 1; a.zero_to = zero_to;
  const int nt = (a.M + 255) / 256;     // every bf16x3 tile form is 256 rows tall (128 x 256 excepted: never picked by the heuristic)
  *tile_rows = 0; *n_tiles = 0;
  if (bn_part != nullptr && (size_t)nt * 3 * d->Cout <= bn_part_floats && pick_b3_tile(zero_to > a.N ? zero_to : a.N) != 3) {
    a.bn_part = bn_part;
    *tile_rows = 256; *n_tiles = nt;
  }
  return run_b3(a, (hipStream_t)stream);
}

// dx (+)= backward-data of a STRIDE-1 convolution from pre-split planes: dy_planes = catseg_split3 of dy (C = Cout rounded up
// to 8 with zero pad), wt_planes = catseg_split3_weight_t of the weights
extern "C" int catseg_conv2d_bwd_data_bf16x3(const catseg_conv_desc* d, const void* dy_planes, const void* wt_planes, float* dx,
                                             int accumulate, catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->stride == 1 && d->kh * d->kw <= 32, "conv bwd_data bf16x3: stride 1, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(dy_planes) && cs_aligned16(wt_planes) && cs_aligned16(dx), "conv bwd_data bf16x3: alignment");
  const int cop = (d->Cout + 7) & ~7;
  CS_REQUIRE((long long)d->B * d->Ho * d->Wo * cop < (1ll << 31) && (long long)d->Cin * d->kh * d->kw * cop < (1ll << 31),
             "conv bwd_data bf16x3: 32-bit offsets");
  B3Args a = {};
  a.a = (const u16*)dy_planes; a.lda = cop; a.a_plane = (long long)d->B * d->Ho * d->Wo * cop;
  a.w = (const u16*)wt_planes; a.ldw = d->kh * d->kw * cop; a.w_plane = (long long)d->Cin * a.ldw;
  a.C = dx; a.ldc = d->ldx; a.bias = nullptr;
  a.M = d->B * d->H * d->W; a.N = d->Cin; a.Cin = cop; a.taps = d->kh * d->kw;
  a.H = d->Ho; a.W = d->Wo; a.Ho = d->H; a.Wo = d->W; a.kw = d->kw; a.stride = 1; a.pad = d->pad; a.dil = d->dil;
  a.sign = -1; a.accumulate = accumulate;
  return run_b3(a, (hipStream_t)stream);
}

// ---- blocked operand planes (the 256 x 256 register-pipelined kernel reads whole cache lines per LDS-DMA instruction) ----------
extern "C" size_t catseg_split3_blocked_elems(long long rows, int C) { return (size_t)3 * (size_t)((C + 15) / 16) * 16 * (size_t)rows; }

// x [rows][ld] fp32 -> blocked planes [3][ceil(C/16)][rows][16]; planar_planes (may be null) additionally receives the layout of
// catseg_split3 ([3][rows][roundup(C, 8)]) from the same pass over x
extern "C" int catseg_split3_blocked(const float* x, long long rows, int C, int ld, void* blocked_planes, void* planar_planes,
                                     catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && ld >= C && ld % 4 == 0 && cs_aligned16(x) && cs_aligned16(blocked_planes) && cs_aligned16(planar_planes),
             "split3_blocked: bad args (ld must be a multiple of 4, pointers 16-byte aligned)");
  const int ldp = (C + 7) & ~7;
  const int c16 = (C + 15) / 16;
  CS_REQUIRE((rows + 63) / 64 < (1ll << 31), "split3_blocked: too many rows");
  hipLaunchKernelGGL(split3_blocked_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)((c16 * 16 + 127) / 128)), dim3(256), 0, (hipStream_t)stream, x,
                     ld, rows, C, ldp, (u16*)blocked_planes, (long long)c16 * rows * 16, (u16*)planar_planes, rows * ldp);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// OHWI weights viewed as [Cout][K = taps * Cin] -> blocked planes [3][K/16][Cout][16] (Cin % 16 == 0): B operand of the forward conv
extern "C" int catseg_split3_weight_blocked(const float* w, int O, int taps, int Cin, void* planes, catseg_stream_t stream) {
  CS_REQUIRE(O > 0 && taps > 0 && Cin > 0 && Cin % 16 == 0 && cs_aligned16(planes), "split3_weight_blocked: needs Cin % 16 == 0");
  const int K = taps * Cin;
  const long long n = (long long)O * K;
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split3_weight_blocked_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, O, K, O, O, taps, Cin, (u16*)planes, n);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// OHWI weights -> blocked planes of the transposed filter bank [3][taps * Opad / 16][Cin][16], Opad = roundup(O, 16): B operand of
// backward-data
extern "C" int catseg_split3_weight_t_blocked(const float* w, int O, int taps, int Cin, void* planes, catseg_stream_t stream) {
  CS_REQUIRE(O > 0 && taps > 0 && Cin > 0 && cs_aligned16(planes), "split3_weight_t_blocked: bad args");
  const int Opad = (O + 15) & ~15;
  const int K = taps * Opad;
  const long long n = (long long)Cin * K;
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split3_weight_blocked_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, Cin, K, O, Opad, taps, Cin, (u16*)planes, n);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// as catseg_conv2d_fwd_bf16x3 / _bnstats with BLOCKED planes: x_planes = catseg_split3_blocked of x (C = Cin, Cin % 16 == 0),
// w_planes = catseg_split3_weight_blocked.  bn_part may be null (then tile_rows / n_tiles are not touched).
extern "C" int catseg_conv2d_fwd_bf16x3_blocked(const catseg_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                                                float* y, int zero_to, float* bn_part, size_t bn_part_floats, int* tile_rows, int* n_tiles,
                                                catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->Cin % 16 == 0 && d->kh * d->kw <= 32, "conv fwd bf16x3 blocked: needs Cin % 16 == 0, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(w_planes) && cs_aligned16(y) && zero_to <= d->ldy, "conv fwd bf16x3 blocked: alignment");
  B3Args a = {};
  a.blocked = 1;
  a.a_rows = d->B * d->H * d->W; a.w_rows = d->Cout;
  a.a = (const u16*)x_planes; a.lda = d->Cin; a.a_plane = (long long)a.a_rows * d->Cin;
  a.w = (const u16*)w_planes; a.ldw = d->kh * d->kw * d->Cin; a.w_plane = (long long)d->Cout * a.ldw;
  a.C = y; a.ldc = d->ldy; a.bias = bias;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Cin = d->Cin; a.taps = d->kh * d->kw;
  a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.sign = 1; a.zero_to = zero_to;
  if (bn_part != nullptr) {
    CS_REQUIRE(tile_rows && n_tiles, "conv fwd bf16x3 blocked: tile_rows / n_tiles");
    const int nt = (a.M + 255) / 256;
    *tile_rows = 0; *n_tiles = 0;
    if ((size_t)nt * 3 * d->Cout <= bn_part_floats) {
      a.bn_part = bn_part;
      *tile_rows = 256; *n_tiles = nt;
    }
  }
  return run_b3(a, (hipStream_t)stream);
}

// inference: y = act(conv(x, w) + bias (+ residual)) from BLOCKED planes in one kernel (the bf16x3 counterpart of
// catseg_conv2d_fwd_fused; w_planes = catseg_split3_weight_blocked of the BatchNorm-folded weights)
extern "C" int catseg_conv2d_fwd_fused_bf16x3_blocked(const catseg_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                                                      const float* residual, int ldr, int relu, float* y, catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->Cin % 16 == 0 && d->kh * d->kw <= 32, "conv fwd fused bf16x3: needs Cin % 16 == 0, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(w_planes) && cs_aligned16(y) && (residual == nullptr || ldr >= d->Cout), "conv fwd fused bf16x3: bad args");
  B3Args a = {};
  a.blocked = 1;
  a.a_rows = d->B * d->H * d->W; a.w_rows = d->Cout;
  a.a = (const u16*)x_planes; a.lda = d->Cin; a.a_plane = (long long)a.a_rows * d->Cin;
  a.w = (const u16*)w_planes; a.ldw = d->kh * d->kw * d->Cin; a.w_plane = (long long)d->Cout * a.ldw;
  a.C = y; a.ldc = d->ldy; a.bias = bias;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Cin = d->Cin; a.taps = d->kh * d->kw;
  a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.sign = 1;
  a.residual = residual; a.ldr = ldr; a.relu = relu;
  return run_b3(a, (hipStream_t)stream);
}

// as catseg_conv2d_bwd_data_bf16x3 with BLOCKED planes: dy_planes = catseg_split3_blocked of dy (C = Cout; the channel tail up to
// roundup(Cout, 16) is zero), wt_planes = catseg_split3_weight_t_blocked
extern "C" int catseg_conv2d_bwd_data_bf16x3_blocked(const catseg_conv_desc* d, const void* dy_planes, const void* wt_planes, float* dx,
                                                     int accumulate, catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->stride == 1 && d->kh * d->kw <= 32, "conv bwd_data bf16x3 blocked: stride 1, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(dy_planes) && cs_aligned16(wt_planes) && cs_aligned16(dx), "conv bwd_data bf16x3 blocked: alignment");
  const int cop = (d->Cout + 15) & ~15;
  B3Args a = {};
  a.blocked = 1;
  a.a_rows = d->B * d->Ho * d->Wo; a.w_rows = d->Cin;
  a.a = (const u16*)dy_planes; a.lda = cop; a.a_plane = (long long)a.a_rows * cop;
  a.w = (const u16*)wt_planes; a.ldw = d->kh * d->kw * cop; a.w_plane = (long long)d->Cin * a.ldw;
  a.C = dx; a.ldc = d->ldx; a.bias = nullptr;
  a.M = d->B * d->H * d->W; a.N = d->Cin; a.Cin = cop; a.taps = d->kh * d->kw;
  a.H = d->Ho; a.W = d->Wo; a.Ho = d->H; a.Wo = d->W; a.kw = d->kw; a.stride = 1; a.pad = d->pad; a.dil = d->dil;
  a.sign = -1; a.accumulate = accumulate;
  return run_b3(a, (hipStream_t)stream);
}

namespace {
int b3t_splits(int tiles, long long P) {
  int best = 1;
  double best_fill = 0.0;
  for (int sp = 1; sp <= 64; ++sp) {
    if (P / sp < 2048 && sp > 1) break;                 // at least 128 K-steps per block
    const double rounds = (double)tiles * sp / 256.0;
    const double fill = rounds / (double)(long long)(rounds + 0.999999);
    if (fill > best_fill + 0.01) { best_fill = fill; best = sp; }
  }
  return best;
}
}  // namespace

// workspace: split-reduction slabs (only when more than one split is planned)
extern "C" size_t catseg_conv2d_bwd_weight_bf16x3_workspace(const catseg_conv_desc* d) {
  if (!d) return 0;
  const long long P = (long long)d->B * d->Ho * d->Wo;
  const int N = d->kh * d->kw * d->Cin;
  const int tiles = ((d->Cout + 255) / 256) * ((N + 255) / 256);
  const int sp = b3t_splits(tiles, P);
  return sp > 1 ? cs_align_up((size_t)sp * d->Cout * N * 4, 256) : 0;
}

// dw[o][ky][kx][c] = sum_p dy[p][o] x[pix(p,ky,kx)][c] from pre-split planes: dy_planes = catseg_split3 of dy (C = Cout),
// x_planes = catseg_split3 of x (C = Cin, Cin % 8 == 0)
extern "C" int catseg_conv2d_bwd_weight_bf16x3(const catseg_conv_desc* d, const void* x_planes, const void* dy_planes, float* dw,
                                               void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->Cin % 8 == 0, "conv bwd_weight bf16x3: dense, Cin % 8 == 0");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(dy_planes) && cs_aligned16(dw), "conv bwd_weight bf16x3: alignment");
  CS_REQUIRE((long long)d->B * d->H * d->W * d->Cin < (1ll << 31), "conv bwd_weight bf16x3: 32-bit offsets");
  const size_t need = catseg_conv2d_bwd_weight_bf16x3_workspace(d);
  if (workspace_bytes < need || (need && !workspace)) {
    catseg_set_error("conv bwd_weight bf16x3: workspace %zu < %zu", workspace_bytes, need);
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  B3TArgs a = {};
  a.P = d->B * d->Ho * d->Wo;
  a.M = d->Cout; a.Cin = d->Cin; a.taps = d->kh * d->kw; a.N = a.taps * d->Cin;
  a.ldo = (d->Cout + 7) & ~7; a.dy = (const u16*)dy_planes; a.dy_plane = (long long)a.P * a.ldo;
  a.ldx = d->Cin; a.x = (const u16*)x_planes; a.x_plane = (long long)d->B * d->H * d->W * d->Cin;
  a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  const int img = d->Ho * d->Wo;
  a.step_b = 16 / img; a.step_qy = (16 % img) / d->Wo; a.step_rx = (16 % img) % d->Wo;
  a.tilesM = (a.M + 255) / 256; a.tilesN = (a.N + 255) / 256;
  const int sp0 = b3t_splits(a.tilesM * a.tilesN, a.P);
  a.rows_per_split = (int)((((long long)a.P + sp0 - 1) / sp0 + 15) / 16 * 16);
  const int sp = (a.P + a.rows_per_split - 1) / a.rows_per_split;
  a.ldc = a.N; a.c_split_stride = (long long)a.M * a.N;
  a.C = sp > 1 ? (float*)workspace : dw;
  a.zero = zero_page_b3();
  hipLaunchKernelGGL(igemm_b3t_kernel, dim3(a.tilesM * a.tilesN * sp), dim3(256), 0, st, a);
  CS_LAUNCH_CHECK();
  if (sp > 1) {
    const long long n4 = (long long)a.M * a.N / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(b3_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n4, sp, n4);
    CS_LAUNCH_CHECK();
  }
  return CATSEG_OK;
}
