// Split-precision implicit-GEMM convolution on the fp16 matrix cores of gfx950: TWO planes, THREE products.
//
// igemm_bf16x3.hip reproduces an fp32 product from three bf16 planes per operand and six MFMA products; under that load the chip is
// power-limited (1.54 - 1.67 GHz) and the large head layers sit at 0.85 of what the sustained clock allows -- the matrix work itself is
// the floor.  fp16 carries 11 significant bits per piece instead of 8:
//   xs = x * 2^e (exact; e chosen per TENSOR so that max|xs| is in [2^14, 2^15)),  h = fp16(xs),  l = fp16(xs - h)
//   xs = h + l + r,  |r| <= 2^-22 |xs|   (for |xs| >= 2^-3; smaller elements lose relative, not absolute precision: |r| <= 2^-25,
//                                          i.e. 2^-39 of the tensor's largest element)
// and a product of two such operands is  hh + hl + lh  (+ ll + ..., relative 2^-22) -- three v_mfma_f32_32x32x16_f16 per 32x32x16
// block, all into ONE fp32 accumulator; every fp16 x fp16 product is exact in fp32.  Representation error 2^-22 per operand (a fourth
// of an fp32 ulp of the operand is lost, about one ulp of the result), accumulation in fp32 as before.  What fp16 needs and bf16 does
// not is the range: 5 exponent bits, so every operand tensor is scaled by a power of two derived from its amax (gradients are ~1e-6)
// and the epilogue scales the result back by 2^-(e_a + e_w).
//
// This file: amax, the two-plane split of activations (blocked and / or planar layout) and weights, igemm_h2w_kernel -- the 256 x 256
// register-pipelined forward / backward-data kernel of igemm_bf16x3.hip (blocked planes, LDS-DMA through buffer resources, one wave
// per SIMD) with 48 MFMAs per K-step in the order  lh | hh | hl  and four 32 KB LDS slots -- and igemm_h2t_kernel, its backward-weight
// counterpart (planar planes, transposed LDS reads).
#include "common.h"

// waves per block of the forward / backward-data kernel: 8 = igemm_h2w8_kernel (two waves per SIMD, the default since round 6),
// 4 = igemm_h2w_kernel (one 512-register wave per SIMD); bit-identical results
int catseg_g_h2w_waves = 8;
int catseg_g_h2w_slow_epilogue = 0;     // (tests: 1 = the general epilogue for every launch of igemm_h2w8_kernel -- bit-identity of the fast one)
extern "C" int catseg_debug_set_h2w_slow_epilogue(int on) {
  catseg_g_h2w_slow_epilogue = on ? 1 : 0;
  return CATSEG_OK;
}
extern "C" int catseg_debug_set_h2w_waves(int waves) {
  catseg_g_h2w_waves = waves == 4 ? 4 : 8;
  return CATSEG_OK;
}

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// ---- amax -----------------------------------------------------------------------------------------------------------------------
// bits of max |x| over [rows][C] (row stride ld), atomicMax on the unsigned bit pattern (non-negative floats order like their bits;
// a NaN is larger than everything and is kept: the split then falls back to e = 0 and NaNs propagate as they would in fp32)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int ld, long long rows, int C, unsigned* __restrict__ out) {
  const int c4n = (C + 3) >> 2;
  const long long n = rows * c4n;
  unsigned m = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const float* src = x + r * ld + c;
    if (c + 3 < C) {
      const f32x4 v = *(const f32x4*)src;
#pragma unroll
      for (int j = 0; j < 4; ++j) m = max(m, __float_as_uint(v[j]) & 0x7FFFFFFFu);
    } else {
      for (int j = 0; c + j < C; ++j) m = max(m, __float_as_uint(src[j]) & 0x7FFFFFFFu);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  __shared__ unsigned sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    if (m) atomicMax(out, m);
  }
}

// exponent e of the prescale 2^e: amax * 2^e in [2^14, 2^15) (fp16's largest finite value is 65504); 0 for an all-zero tensor, for
// denormal amax and for inf / NaN; clamped so that 2^e and 2^-e stay normal floats
__device__ __forceinline__ int h2_exponent(unsigned amax_bits) {
  const int ex = (int)((amax_bits >> 23) & 0xFF);
  if (ex == 0 || ex == 255) return 0;
  const int e = 14 - (ex - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

__device__ __forceinline__ void split2h_one(float v, int e, u16& h, u16& l) {
  const float xs = __builtin_ldexpf(v, e);
  const _Float16 hh = (_Float16)xs;
  const float r = xs - (float)hh;
  h = __builtin_bit_cast(u16, hh);
  l = __builtin_bit_cast(u16, (_Float16)r);
}

// fp32 [rows][ld] -> two fp16 planes, prescaled by 2^e (e from amax_bits, written to *e_out), in the BLOCKED layout [C16][rows][16]
// (C16 = ceil(C / 16), channel tail zero; operand of igemm_h2w_kernel) and / or the planar layout [rows][ldp], ldp = roundup(C, 8)
// (operand of igemm_h2t_kernel), from ONE pass over x.  Block = 64 rows x 128 channels through LDS, as split3_blocked_kernel.
__global__ __launch_bounds__(256) void split2h_kernel(const float* __restrict__ x, int ld, long long rows, int C, int ldp,
                                                      const unsigned* __restrict__ amax_bits, u16* __restrict__ blk, long long blk_plane,
                                                      u16* __restrict__ planar, long long planar_plane, int* __restrict__ e_out) {
  constexpr int RS = 128 + 8;
  __shared__ __attribute__((aligned(16))) u16 sh[2][64 * RS];
  const int e = h2_exponent(*amax_bits);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *e_out = e;
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
    u16 h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2h_one(v[j], e, h[j], l[j]);
    u16* d = &sh[0][row * RS + c4];
    *(unsigned long long*)d = (unsigned long long)h[0] | ((unsigned long long)h[1] << 16) | ((unsigned long long)h[2] << 32) | ((unsigned long long)h[3] << 48);
    *(unsigned long long*)(d + 64 * RS) = (unsigned long long)l[0] | ((unsigned long long)l[1] << 16) | ((unsigned long long)l[2] << 32) | ((unsigned long long)l[3] << 48);
  }
  __syncthreads();
  if (planar != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + t, row = q >> 4, c8 = (q & 15) * 8;
      if (r0 + row < rows && c0 + c8 < ldp) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          *(half8*)(planar + pl * planar_plane + (r0 + row) * ldp + c0 + c8) = *(const half8*)&sh[pl][row * RS + c8];
      }
    }
  }
  if (blk == nullptr) return;
  const int c16 = (C + 15) >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + t, cc = q >> 7, row = (q & 127) >> 1, half = (q & 1) * 8;
    const int chunk = (c0 >> 4) + cc;
    if (r0 + row < rows && chunk < c16) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        *(half8*)(blk + pl * blk_plane + ((long long)chunk * rows + r0 + row) * 16 + half) = *(const half8*)&sh[pl][row * RS + cc * 16 + half];
    }
  }
}

// ---- the channel concatenation of up to four bilinearly resized NHWC tensors, written ONLY as blocked planes ------------------------------
// HRNet's head input (models/HRNetv2.py:505-508 of the reference: F.interpolate(x_i, size of x_0, mode='bilinear', align_corners=False) of
// branches 1..3, torch.cat with branch 0) feeds nothing but the f16x2 kernels of the two 3 x 3 head convolutions: the fp32 concatenation
// (752 MB at 8 x 136 x 240 x 720), the copy of branch 0 into it and the split pass that read it back are replaced by ONE launch that
// interpolates, splits and writes the planes.  Thread = (output pixel, 16-channel chunk): a wave writes 64 pixels x 32 bytes = 2 KB
// contiguous per plane.  The interpolation is bilinear_fwd_kernel's expression (csrc/pointwise.hip; ATen's source index in fp32);
// a source of the output's size is copied.  Exponent from amax_bits (the maximum over the sources' records: interpolation weights lie in
// [0, 1] and sum to 1), as split2h_kernel.
struct CatSrc { const float* x; int ld, H, W, chunk0; float sh, sw; };
struct CatArgs { CatSrc s[4]; int nsrc, nchunks, B, Ho, Wo; long long rows; };

__device__ __forceinline__ void cat_lerp(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float sidx = scale * (dst + 0.5f) - 0.5f;        // area_pixel_compute_source_index, align_corners = False
  sidx = sidx < 0.f ? 0.f : sidx;
  i0 = (int)sidx;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = sidx - i0;
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void concat_bilinear_split2h_kernel(const CatArgs a, const unsigned* __restrict__ amax_bits, u16* __restrict__ blk,
                                                                      long long blk_plane, int* __restrict__ e_out) {
  const int e = h2_exponent(*amax_bits);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *e_out = e;
  const int chunk = blockIdx.y;
  int si = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) si = (k < a.nsrc && chunk >= a.s[k].chunk0) ? k : si;
  const CatSrc s = a.s[si];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.rows) return;
  const int c = (chunk - s.chunk0) * 16;
  const int ox = (int)(p % a.Wo);
  const long long t = p / a.Wo;
  const int oy = (int)(t % a.Ho), b = (int)(t / a.Ho);
  f32x4 v[4];
  if (s.H == a.Ho && s.W == a.Wo) {
    const float* r = s.x + (((long long)b * s.H + oy) * s.W + ox) * s.ld + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *(const f32x4*)(r + 4 * q);
  } else {
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    cat_lerp(s.sh, oy, s.H, y0, y1, ly0, ly1);
    cat_lerp(s.sw, ox, s.W, x0, x1, lx0, lx1);
    const float* r0 = s.x + ((long long)b * s.H + y0) * s.W * s.ld + c;
    const float* r1 = s.x + ((long long)b * s.H + y1) * s.W * s.ld + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 a0 = *(const f32x4*)(r0 + x0 * s.ld + 4 * q), a1 = *(const f32x4*)(r0 + x1 * s.ld + 4 * q);
      const f32x4 b0 = *(const f32x4*)(r1 + x0 * s.ld + 4 * q), b1 = *(const f32x4*)(r1 + x1 * s.ld + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[q][j] = ly0 * (lx0 * a0[j] + lx1 * a1[j]) + ly1 * (lx0 * b0[j] + lx1 * b1[j]);
    }
  }
  u16 h[16], l[16];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) split2h_one(v[q][j], e, h[4 * q + j], l[4 * q + j]);
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  u16* d = blk + ((long long)chunk * a.rows + p) * 16;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    u16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) { hv[j] = h[8 * half + j]; lv[j] = l[8 * half + j]; }
    *(u16x8*)(d + 8 * half) = hv;
    *(u16x8*)(d + blk_plane + 8 * half) = lv;
  }
}

// fp32 weights viewed as [N][K] -> blocked fp16 planes [K/16][N][16], prescaled; T = the transposed filter bank of backward-data
// (row n = input channel c, k = tap * Opad + o, source w[(o * taps + tap) * Cin + c], zero for o >= O)
template <bool T>
__global__ __launch_bounds__(256) void split2h_weight_blocked_kernel(const float* __restrict__ w, int N, int K, int O, int Opad, int taps, int Cin,
                                                                     const unsigned* __restrict__ amax_bits, u16* __restrict__ out, long long plane,
                                                                     int* __restrict__ e_out) {
  const int e = h2_exponent(*amax_bits);
  if (blockIdx.x == 0 && threadIdx.x == 0) *e_out = e;
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
    u16 h, l;
    split2h_one(v, e, h, l);
    out[i] = h;
    out[i + plane] = l;
  }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------------
struct H2Args {
  const u16* a;       // activation planes, blocked [2][Cin/16][a_rows][16]
  long long a_plane;  // elements between planes
  const u16* w;       // weight planes, blocked [2][taps * Cin / 16][w_rows][16]
  long long w_plane;
  const int* ea;      // device: prescale exponents of the two operands
  const int* ew;
  float* C;
  int ldc;
  const float* bias;
  int M, N, Cin, taps;
  int H, W, Ho, Wo;   // H, W: source image of the gather; Ho, Wo: the grid the GEMM rows decode over
  int kw, stride, pad, dil;
  int sign;           // +1: forward gather; -1: stride-1 backward-data gather
  int tilesM, tilesN;
  int zero_to, accumulate;
  float* bn_part;     // per (M-tile, channel) BatchNorm partials [tilesM][3][N], or nullptr
  int a_rows, w_rows;
  const float* residual;   // fused inference epilogue: v = act(v + bias (+ residual))
  int ldr, relu;
};

constexpr int h2_waitcnt(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

// 4 waves (one per SIMD), block tile 256 x 256, wave tile 128 x 128 (4 x 4 MFMA tiles of 32 x 32), K-step = 16.
// Per K-step 48 MFMAs (1536 cycles): products in the order  lh | hh | hl.  An iteration of the K loop is  hh(k), hl(k), lh(k+1):
// every fragment set is re-read for step k+1 right behind its last use in step k --
//   Al behind lh(k) [read under hh(k)],  Bh behind hh(k) [under hl(k)],  Ah and Bl behind hl(k) [under lh(k+1), which needs Al, Bh] --
// so all fragment reads of an iteration come from ONE slot (k+1), nothing is double buffered (64 fragment registers + 256
// accumulators) and each read has >= 8 MFMAs (256 cycles) before its first use.
// LDS: four 32 KB slots.  Iteration k reads slot (k+1) % 4 while steps k+2, k+3 are landing and the LDS-DMA of step k+4 is issued
// into slot k % 4 (last read in iteration k-1: the barrier at the top of the iteration separates them).  One barrier per K-step;
// a load has three K-steps (4608 cycles) of latency budget.
__global__ __launch_bounds__(256, 1) void igemm_h2w_kernel(const H2Args p) {
  constexpr int TM = 4, TN = 4, WGN = 2;
  constexpr int BM = 256, BN = 256;
  constexpr int PLANE_A = BM * 32, PLANE_B = BN * 32;
  constexpr int SLAB = 2 * (PLANE_A + PLANE_B);
  constexpr int NSLOT = 4;
  constexpr int NA = 2, NB = 2;   // 16-byte chunks per thread per plane
  __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLAB];

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
      aoff[j] = (b * p.H + y0) * p.W + x0;   // pixel index of tap (0, 0)
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

  const int nck = (p.Cin + 15) >> 4;
  const int nks = p.taps * nck;
  int ttap = 0, tky = 0, tkx = 0, tck = 0;
  // LDS-DMA sources through two raw buffer resources (one per operand, both planes inside; the host checks 2 planes < 4 GB): a lane's
  // source = 32-bit BYTE offset; padding taps, channel tails and rows past M / N get an out-of-range offset = zeros in LDS
  const unsigned apl_b = (unsigned)(p.a_plane * 2), wpl_b = (unsigned)(p.w_plane * 2);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, (short)0, (int)(2u * apl_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)(2u * wpl_b), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned VA[NA], VB[NB];   // byte offsets of one K-step's pieces inside plane 0 (the plane offset rides in the instruction's soffset)
  auto prepA = [&](const int j) {
    const bool ok = (int)(ttap < p.taps) & (int)((amask[j] >> (ttap & 31)) & 1u) & (int)((tck * 16 + achunk[j]) < p.Cin);
    const unsigned v = (unsigned)((tck * p.a_rows + aoff[j] + p.sign * (tky * p.dil * p.W + tkx * p.dil)) * 16 + achunk[j]) * 2u;
    VA[j] = ok ? v : OOB;
  };
  auto prepB = [&](const int j) {
    const bool ok = (int)(ttap < p.taps) & (int)(brow[j] < p.N) & (int)((tck * 16 + bchunk[j]) < p.Cin);
    const unsigned v = (unsigned)(((ttap * nck + tck) * p.w_rows + brow[j]) * 16 + bchunk[j]) * 2u;   // [K/16][N][16]
    VB[j] = ok ? v : OOB;
  };
  // K order: 16-channel chunk outer, filter taps inner (the nine taps of a chunk re-read the same contiguous run of a blocked plane
  // shifted by a few pixels: L1 / L2 hits instead of nine passes over the whole plane through the fabric)
  auto advance = [&]() {
    const int nx = tkx + 1, nt = ttap + 1;
    const bool wrapx = nx == p.kw, wrapt = nt == p.taps;
    tkx = wrapx ? 0 : nx;
    tky = wrapt ? 0 : (wrapx ? tky + 1 : tky);
    ttap = wrapt ? 0 : nt;
    tck += wrapt ? 1 : 0;
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
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(s + 2 * PLANE_A + pl * PLANE_B + ((w - 2) * 256 + wave * 64) * 16),
                                               16, VB[w - 2], pl * wpl_b, 0, 0);
  };
  auto issue = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(buf, i);
  };

  half8 Ah[TM], Bh[TN], Al[TM], Bl[TN];
  int ra0, rb0;
  {
    const int rowa = wm * 128 + l31, rowb = wn * 128 + l31;   // (row + 32 t keeps (row >> 3) & 1: one swizzle per lane)
    ra0 = rowa * 32 + ((h ^ ((rowa >> 3) & 1)) << 4);
    rb0 = 2 * PLANE_A + rowb * 32 + ((h ^ ((rowb >> 3) & 1)) << 4);
  }
#define H2_READ(dst, base, off) dst = *(const half8*)((base) + (off))
#define H2_MFMA(A_, t_, B_, u_)                                                                              \
  do {                                                                                                        \
    acc[t_][u_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[t_], B_[u_], acc[t_][u_], 0, 0, 0);                \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)
#define H2_SYNC()                                        \
  do {                                                   \
    __builtin_amdgcn_s_waitcnt(h2_waitcnt(16, 0));       \
    __builtin_amdgcn_s_barrier();                        \
  } while (0)

  if (nks > 0) {
    prep();
    issue(0);
    prep();
    issue(1);                       // (unconditional: past the end of the reduction every offset is out of range = zeros)
    prep();
    issue(2);
    prep();
    issue(3);
    prep();
    __builtin_amdgcn_s_waitcnt(h2_waitcnt(24, 0));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      const char* aa_ = smem + ra0;
      const char* bb_ = smem + rb0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        H2_READ(Al[t], aa_, 1 * PLANE_A + t * 1024);
        H2_READ(Bh[t], bb_, 0 * PLANE_B + t * 1024);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        H2_READ(Ah[t], aa_, 0 * PLANE_A + t * 1024);
        H2_READ(Bl[t], bb_, 1 * PLANE_B + t * 1024);
      }
    }
    // lh(0)
    H2_MFMA(Al, 0, Bh, 0); H2_MFMA(Al, 0, Bh, 1); H2_MFMA(Al, 0, Bh, 2); H2_MFMA(Al, 0, Bh, 3);
    H2_MFMA(Al, 1, Bh, 0); H2_MFMA(Al, 1, Bh, 1); H2_MFMA(Al, 1, Bh, 2); H2_MFMA(Al, 1, Bh, 3);
    H2_MFMA(Al, 2, Bh, 0); H2_MFMA(Al, 2, Bh, 1); H2_MFMA(Al, 2, Bh, 2); H2_MFMA(Al, 2, Bh, 3);
    H2_MFMA(Al, 3, Bh, 0); H2_MFMA(Al, 3, Bh, 1); H2_MFMA(Al, 3, Bh, 2); H2_MFMA(Al, 3, Bh, 3);
    int nxt = 1, fill = 0;
    for (int k = 0; k < nks; ++k) {
      // all but my newest 16 LDS-DMA pieces (steps k+2, k+3) have landed: step k+1 is there; every fragment read has returned; the
      // barrier: ... everybody's, and every wave has read the fragments of step k out of slot k % 4, which step k+4 now overwrites
      H2_SYNC();
      asm volatile("" ::: "memory");
      const char* aa_ = smem + nxt * SLAB + ra0;
      const char* bb_ = smem + nxt * SLAB + rb0;
      // hh(k); Al(k+1) read; the eight pieces of step k+4
      H2_READ(Al[0], aa_, PLANE_A + 0 * 1024); H2_MFMA(Ah, 0, Bh, 0);
      H2_READ(Al[1], aa_, PLANE_A + 1 * 1024); H2_MFMA(Ah, 0, Bh, 1);
      H2_READ(Al[2], aa_, PLANE_A + 2 * 1024); H2_MFMA(Ah, 0, Bh, 2);
      H2_READ(Al[3], aa_, PLANE_A + 3 * 1024); H2_MFMA(Ah, 0, Bh, 3);
      piece(fill, 0); H2_MFMA(Ah, 1, Bh, 0);
      piece(fill, 1); H2_MFMA(Ah, 1, Bh, 1);
      piece(fill, 2); H2_MFMA(Ah, 1, Bh, 2);
      piece(fill, 3); H2_MFMA(Ah, 1, Bh, 3);
      piece(fill, 4); H2_MFMA(Ah, 2, Bh, 0);
      piece(fill, 5); H2_MFMA(Ah, 2, Bh, 1);
      piece(fill, 6); H2_MFMA(Ah, 2, Bh, 2);
      piece(fill, 7); H2_MFMA(Ah, 2, Bh, 3);
      H2_MFMA(Ah, 3, Bh, 0); H2_MFMA(Ah, 3, Bh, 1); H2_MFMA(Ah, 3, Bh, 2); H2_MFMA(Ah, 3, Bh, 3);
      // hl(k); Bh(k+1) read; the addresses of step k+5
      H2_READ(Bh[0], bb_, 0 * 1024); H2_MFMA(Ah, 0, Bl, 0);
      H2_READ(Bh[1], bb_, 1 * 1024); H2_MFMA(Ah, 0, Bl, 1);
      H2_READ(Bh[2], bb_, 2 * 1024); H2_MFMA(Ah, 0, Bl, 2);
      H2_READ(Bh[3], bb_, 3 * 1024); H2_MFMA(Ah, 0, Bl, 3);
      H2_MFMA(Ah, 1, Bl, 0);
      prepA(0); H2_MFMA(Ah, 1, Bl, 1);
      H2_MFMA(Ah, 1, Bl, 2);
      prepA(1); H2_MFMA(Ah, 1, Bl, 3);
      H2_MFMA(Ah, 2, Bl, 0);
      prepB(0); H2_MFMA(Ah, 2, Bl, 1);
      H2_MFMA(Ah, 2, Bl, 2);
      prepB(1); H2_MFMA(Ah, 2, Bl, 3);
      H2_MFMA(Ah, 3, Bl, 0);
      advance(); H2_MFMA(Ah, 3, Bl, 1);
      H2_MFMA(Ah, 3, Bl, 2); H2_MFMA(Ah, 3, Bl, 3);
      // lh(k+1) (step nks is all zeros); Ah(k+1), Bl(k+1) read
      H2_READ(Ah[0], aa_, 0 * 1024); H2_MFMA(Al, 0, Bh, 0);
      H2_READ(Ah[1], aa_, 1 * 1024); H2_MFMA(Al, 0, Bh, 1);
      H2_READ(Ah[2], aa_, 2 * 1024); H2_MFMA(Al, 0, Bh, 2);
      H2_READ(Ah[3], aa_, 3 * 1024); H2_MFMA(Al, 0, Bh, 3);
      H2_READ(Bl[0], bb_, PLANE_B + 0 * 1024); H2_MFMA(Al, 1, Bh, 0);
      H2_READ(Bl[1], bb_, PLANE_B + 1 * 1024); H2_MFMA(Al, 1, Bh, 1);
      H2_READ(Bl[2], bb_, PLANE_B + 2 * 1024); H2_MFMA(Al, 1, Bh, 2);
      H2_READ(Bl[3], bb_, PLANE_B + 3 * 1024); H2_MFMA(Al, 1, Bh, 3);
      H2_MFMA(Al, 2, Bh, 0); H2_MFMA(Al, 2, Bh, 1); H2_MFMA(Al, 2, Bh, 2); H2_MFMA(Al, 2, Bh, 3);
      H2_MFMA(Al, 3, Bh, 0); H2_MFMA(Al, 3, Bh, 1); H2_MFMA(Al, 3, Bh, 2); H2_MFMA(Al, 3, Bh, 3);
      nxt = (nxt + 1) & 3;
      fill = (fill + 1) & 3;
    }
  }
#undef H2_READ
#undef H2_MFMA
#undef H2_SYNC

  // back to the operands' scale: 2^-(e_a + e_w) in two exact steps (each exponent is within [-100, 100])
  {
    const int ea = -*p.ea, ew = -*p.ew;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = __builtin_ldexpf(__builtin_ldexpf(acc[t][u][r], ea), ew);
  }

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
      // what is added to the tile (previous contents when accumulating, the residual branch) is read for all 16 rows FIRST: loads
      // interleaved with the stores cannot be hoisted over them (they may alias) and each waited for its own round trip -- a second
      // head's backward-data accumulating into the shared 720-channel gradient cost 5.9 instead of 4.4 ms
      float add[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float a = 0.f;
        if (row < p.M && col < p.N) {
          if (p.accumulate) a = p.C[(long long)row * p.ldc + col];
          if (p.residual != nullptr) a += p.residual[(long long)row * p.ldr + col];
        }
        add[r] = a;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < p.M) {
          float* dst = p.C + (long long)row * p.ldc + col;
          if (col < p.N) {
            float v = (acc[t][u][r] + bv) + add[r];
            if (p.relu) v = fmaxf(v, 0.f);
            *dst = v;
          } else if (col < p.zero_to) {
            *dst = 0.f;
          }
        }
      }
    }
}


// ---- the same GEMM with TWO waves per SIMD (round 6) ------------------------------------------------------------------------------------
// igemm_h2w_kernel runs one 512-register wave per SIMD: the wave that issues the MFMAs also issues the step's eight LDS-DMA pieces, and an
// LDS-DMA instruction holds the wave's issue slot for ~60 - 180 cycles (MI355X_MICROARCH.md, cycle constants) against the 32 cycles of the MFMA
// executing beside it -- the matrix pipe idles behind every piece: the kernel ran at 0.67 - 0.80 of what a bare MFMA loop sustains on the same
// chip (tools/probe/mfma_shape_probe.hip: 1625 TFLOP/s of fp16 MFMA = 542 fp32-equivalent at the clock the chip holds under this load).
// Here a block is 512 threads = 8 waves, two per SIMD, each with a 128 x 64 wave tile (4 x 2 MFMA tiles, 128 accumulator + 48 fragment
// registers: <= 256 in all), still ONE 256 x 256 block tile per CU and the same four 32 KB LDS slots: while one wave of a SIMD issues its four
// pieces or waits for its fragment reads, its partner's MFMAs keep the pipe busy -- the hardware interleaves, no hand-placed schedule.
// Per wave and K-step: 24 MFMAs, 12 ds_read_b128, 4 LDS-DMA pieces, one barrier.  Every output element accumulates the products of a K-step
// in the order lh, hh, hl and the K-steps in the order of igemm_h2w_kernel: results (and BatchNorm partials) are BIT-IDENTICAL to it.
template <bool FAST>
__global__ __launch_bounds__(512, 2) void igemm_h2w8_kernel(const H2Args p) {
  constexpr int TM = 4, TN = 2;
  constexpr int BM = 256, BN = 256;
  constexpr int PLANE_A = BM * 32, PLANE_B = BN * 32;
  constexpr int SLAB = 2 * (PLANE_A + PLANE_B);
  constexpr int NSLOT = 4;
  __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLAB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
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

  // staging: thread q = tid owns ONE 16-byte chunk per plane and operand: row q >> 1, LDS half q & 1 (lane-linear LDS-DMA), logical
  // 8-channel chunk = half ^ ((row >> 3) & 1) -- the image of igemm_h2w_kernel
  int aoff = 0, achunk, bchunk, brow;
  unsigned amask = 0;
  {
    const int row = tid >> 1;
    achunk = ((tid & 1) ^ ((row >> 3) & 1)) * 8;
    bchunk = achunk;
    brow = n0 + row;
    const int r = m0 + row;
    if (r < p.M) {
      const int hw = p.Ho * p.Wo;
      const int b = r / hw, rem = r - b * hw;
      const int y = rem / p.Wo, x = rem - y * p.Wo;
      const int y0 = p.sign > 0 ? y * p.stride - p.pad : y + p.pad;
      const int x0 = p.sign > 0 ? x * p.stride - p.pad : x + p.pad;
      aoff = (b * p.H + y0) * p.W + x0;   // pixel index of tap (0, 0)
      const int kh = p.taps / p.kw;
      int t = 0;
      for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < p.kw; ++kx, ++t) {
          const int yy = y0 + p.sign * ky * p.dil, xx = x0 + p.sign * kx * p.dil;
          if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) amask |= 1u << t;
        }
    }
  }
  const int nck = (p.Cin + 15) >> 4;
  const int nks = p.taps * nck;
  int ttap = 0, tky = 0, tkx = 0, tck = 0;
  const unsigned apl_b = (unsigned)(p.a_plane * 2), wpl_b = (unsigned)(p.w_plane * 2);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, (short)0, (int)(2u * apl_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)(2u * wpl_b), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  unsigned VA, VB;
  auto prep = [&]() {
    const bool oka = (int)(ttap < p.taps) & (int)((amask >> (ttap & 31)) & 1u) & (int)((tck * 16 + achunk) < p.Cin);
    const unsigned va = (unsigned)((tck * p.a_rows + aoff + p.sign * (tky * p.dil * p.W + tkx * p.dil)) * 16 + achunk) * 2u;
    VA = oka ? va : OOB;
    const bool okb = (int)(ttap < p.taps) & (int)(brow < p.N) & (int)((tck * 16 + bchunk) < p.Cin);
    const unsigned vb = (unsigned)(((ttap * nck + tck) * p.w_rows + brow) * 16 + bchunk) * 2u;   // [K/16][N][16]
    VB = okb ? vb : OOB;
    const int nx = tkx + 1, nt = ttap + 1;
    const bool wrapx = nx == p.kw, wrapt = nt == p.taps;
    tkx = wrapx ? 0 : nx;
    tky = wrapt ? 0 : (wrapx ? tky + 1 : tky);
    ttap = wrapt ? 0 : nt;
    tck += wrapt ? 1 : 0;
  };
  auto issue = [&](const int buf) {
    char* s = smem + buf * SLAB + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(s), 16, VA, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(s + PLANE_A), 16, VA, apl_b, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(s + 2 * PLANE_A), 16, VB, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(s + 2 * PLANE_A + PLANE_B), 16, VB, wpl_b, 0, 0);
  };

  int ra0, rb0;
  {
    const int rowa = wm * 128 + l31, rowb = wn * 64 + l31;   // (row + 32 t keeps (row >> 3) & 1: one swizzle per lane)
    ra0 = rowa * 32 + ((h ^ ((rowa >> 3) & 1)) << 4);
    rb0 = 2 * PLANE_A + rowb * 32 + ((h ^ ((rowb >> 3) & 1)) << 4);
  }

  if (nks > 0) {
    prep(); issue(0);
    prep(); issue(1);              // (unconditional: past the end of the reduction every offset is out of range = zeros)
    prep(); issue(2);
    prep();                        // the addresses of step 3
    int cur = 0, fill = 3;
    for (int k = 0; k < nks; ++k) {
      // all but my newest 8 pieces (steps k+1, k+2) have landed: step k is there -- the barrier: everybody's; and every wave has read the
      // fragments of step k-1 out of slot (k+3) % 4, which step k+3 now overwrites
      __builtin_amdgcn_s_waitcnt(h2_waitcnt(8, 15));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(fill);
      prep();
      const char* aa_ = smem + cur * SLAB + ra0;
      const char* bb_ = smem + cur * SLAB + rb0;
      half8 Ah[TM], Al[TM], Bh[TN], Bl[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) Al[t] = *(const half8*)(aa_ + PLANE_A + t * 1024);
#pragma unroll
      for (int u = 0; u < TN; ++u) Bh[u] = *(const half8*)(bb_ + u * 1024);
#pragma unroll
      for (int t = 0; t < TM; ++t) Ah[t] = *(const half8*)(aa_ + t * 1024);
#pragma unroll
      for (int u = 0; u < TN; ++u) Bl[u] = *(const half8*)(bb_ + PLANE_B + u * 1024);
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[t], Bh[u], acc[t][u], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[t], Bh[u], acc[t][u], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[t], Bl[u], acc[t][u], 0, 0, 0);
      cur = (cur + 1) & 3;
      fill = (fill + 1) & 3;
    }
  }

  // ---- FAST epilogue (round 6; a property of the LAUNCH: M a multiple of 256, no residual, no zero-filled pad columns -- run_h2 decides).  The
  // general epilogue below predicates each of a lane's 128 stores with scalar branches, computes a 64-bit address per element, scales with two
  // ldexps per element and selects the valid rows of the BatchNorm partials: ~3500 instructions per wave and tile with the matrix pipe idle --
  // 40 % of a tile's time at K = 1024 (32 K-steps of 24 MFMAs: the 1 x 1 layers of the ResNet trunks and the OCR bottleneck), 10 % at K = 6480.
  // Here: one ldexp, column validity once per column tile (N = 720 of the HRNet head's backward-data leaves the third column tile ragged),
  // buffer loads / stores with one per-lane offset, the accumulator row in the scalar offset and the column tile in the immediate.  Same
  // operations on every element in the same order ((acc + bias) + previous contents, then ReLU): bit-identical outputs and partials.
  if constexpr (FAST) {
    const int ea_u = __builtin_amdgcn_readfirstlane(*p.ea), ew_u = __builtin_amdgcn_readfirstlane(*p.ew);
    const int es = -(ea_u + ew_u);
    if (es >= -120 && es <= 120) {
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][u][r] = __builtin_ldexpf(acc[t][u][r], es);
    } else {
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][u][r] = __builtin_ldexpf(__builtin_ldexpf(acc[t][u][r], -ea_u), -ew_u);
    }
    bool cok[TN];
    float bv[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int col = n0 + wn * 64 + u * 32 + l31;
      cok[u] = col < p.N;
      bv[u] = (p.bias != nullptr && cok[u]) ? p.bias[col] : 0.f;
    }
    if (p.bn_part != nullptr) {
      __builtin_amdgcn_s_waitcnt(h2_waitcnt(0, 0));      // (the zero pieces issued past the end of the reduction still target the slots)
      __syncthreads();
      int colv[TN];
#pragma unroll
      for (int u = 0; u < TN; ++u) colv[u] = wn * 64 + u * 32 + l31;
      cs_tile_bn_partials<TN, TM * 16, 2, false>(
          (float*)smem, BN, colv, h == 0, wm, BM, [&](int j, int i) { return acc[i >> 4][j][i & 15] + bv[j]; }, [&](int) { return true; },
          p.bn_part + (long long)tile_m * 3 * p.N, p.N, n0);
    }
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long long)m0 * p.ldc + n0), (short)0,
                                                                        (int)(256u * (unsigned)p.ldc * 4u), 0x00020000);
    const int ld4 = p.ldc * 4;
    const int voff = (wm * 128 + 4 * h) * ld4 + (wn * 64 + l31) * 4;
    const bool accum = p.accumulate != 0, relu = p.relu != 0;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        if (cok[u]) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[t][u][r] + bv[u];
          if (accum) {
            float o[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
              o[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsC, voff + 128 * u, (t * 32 + (r & 3) + 8 * (r >> 2)) * ld4, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += o[r];
          }
          if (relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rsC, voff + 128 * u, (t * 32 + (r & 3) + 8 * (r >> 2)) * ld4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }

  // back to the operands' scale: 2^-(e_a + e_w) in two exact steps (each exponent is within [-100, 100])
  {
    const int ea = -*p.ea, ew = -*p.ew;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = __builtin_ldexpf(__builtin_ldexpf(acc[t][u][r], ea), ew);
  }

  if (p.bn_part != nullptr) {   // BatchNorm batch statistics of this tile (common.h: cs_tile_bn_partials)
    __builtin_amdgcn_s_waitcnt(h2_waitcnt(0, 0));      // (the zero pieces issued past the end of the reduction still target the slots)
    __syncthreads();
    int colv[TN];
    float bvv[TN];
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      colv[u] = wn * 64 + u * 32 + l31;
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
      const int col = n0 + wn * 64 + u * 32 + l31;
      const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
      float add[16];                  // (what is added to the tile is read for all 16 rows first: see igemm_h2w_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float a = 0.f;
        if (row < p.M && col < p.N) {
          if (p.accumulate) a = p.C[(long long)row * p.ldc + col];
          if (p.residual != nullptr) a += p.residual[(long long)row * p.ldr + col];
        }
        add[r] = a;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < p.M) {
          float* dst = p.C + (long long)row * p.ldc + col;
          if (col < p.N) {
            float v = (acc[t][u][r] + bv) + add[r];
            if (p.relu) v = fmaxf(v, 0.f);
            *dst = v;
          } else if (col < p.zero_to) {
            *dst = 0.f;
          }
        }
      }
    }
}


// ---- backward-weight: dW[o][tap][c] = sum_p dy[p][o] * x[pix(p, tap)][c] as a "TN" GEMM (M = Cout, N = taps * Cin, reduction over the
// output pixels, split over blocks, partial slabs reduced afterwards): igemm_b3t_kernel of igemm_bf16x3.hip with two fp16 planes.
// Planar planes [pixel][channels]; LDS images [16 pixels][256 columns] per plane, chunks swizzled by (pixel & 3) << 2, fragments by
// the hardware transpose ds_read_b64_tr_b16.  48 MFMAs per K-step in the order lh | hh | hl, iteration = hh(k), hl(k), lh(k+1) with
// every fragment set re-read for step k+1 behind its last use (all from the slot of step k+1); NSLOT slots, LDS-DMA NSLOT - 1 K-steps
// ahead (step k+NSLOT into the slot of step k, behind the barrier that ends its reads).
struct H2TArgs {
  const u16* dy;  long long dy_plane; int ldo;     // [P][ldo] planes, ldo = roundup(Cout, 8)
  const u16* x;   long long x_plane;  int ldx;     // [B*H*W][ldx] planes
  const int* edy; const int* ex;                   // device: prescale exponents
  float* C;       int ldc; long long c_split_stride;
  int M, N, Cin, taps;                             // M = Cout, N = taps * Cin
  int P, rows_per_split;                           // output pixels; multiple of 16 per split
  int H, W, Ho, Wo, kw, stride, pad, dil;
  int step_b, step_qy, step_rx;                    // 16 pixels = step_b images + step_qy rows + step_rx pixels
  int tilesM, tilesN;
  int blocked;                                     // planes in the BLOCKED layout [C/16][rows][16] (the forward / backward-data operand): see below
  long long x_rows;                                // blocked: pixel rows of the x planes (B * H * W)
};

__global__ __launch_bounds__(256, 1) void igemm_h2t_kernel(const H2TArgs p) {
  constexpr int TM = 4, TN = 4;
  constexpr int PLANE = 16 * 256 * 2;             // bytes of one operand plane image: 16 pixel rows x 256 columns of fp16
  constexpr int SLAB = 4 * PLANE;                 // A planes 0..1, then B planes 0..1
#ifndef H2T_NSLOT
#define H2T_NSLOT 4
#endif
  constexpr int NSLOT = H2T_NSLOT;                // LDS-DMA runs NSLOT - 1 K-steps ahead
  __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLAB];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  // one-dimensional grid over (pixel slab, tile): the blocks that share an XCD get a contiguous run of it (all tiles of a slab on one L2)
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

  // staging: chunk q = j * 256 + tid of a plane image: pixel row krow = q >> 5, position q & 31 (lane-linear LDS-DMA), logical
  // 8-column chunk = position ^ ((krow & 3) << 2)
  int arow[2], acol[2];
  int brow[2], tb[2], ty[2], tx[2], bky[2], bkx[2], bch[2];
  bool bcv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#ifdef H2T_STAGE2
    const int q = j * 256 + tid, krow = q >> 5, cl = (q & 31) ^ ((krow & 3) << 2);
#else
    // LDS image of a plane: [group of 4 pixel rows][half of the 256 columns][pixel row & 3][16 chunks] -- one LDS-DMA instruction (64 lanes,
    // 1 KB of LDS) covers FOUR consecutive pixel rows x 16 chunks, so that with blocked planes a 16-channel chunk contributes 4 x 32 = 128
    // contiguous bytes per instruction (two rows x 32 chunks, the round-3 image, made that 64: +4 % on the kernel)
    const int q = j * 256 + tid, krow = ((q >> 7) << 2) | ((q >> 4) & 3), cl = ((((q >> 6) & 1) << 4) | (q & 15)) ^ ((krow & 3) << 2);
#endif
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
  // LDS-DMA sources through two raw buffer resources (one per operand, both planes inside; the host checks 2 planes < 4 GB): a lane's
  // source = 32-bit BYTE offset inside plane 0, the plane offset rides in the instruction's soffset; rows past the slab, padding taps
  // and column tails get an out-of-range offset = zeros in LDS.  (With global_load_lds + 64-bit pointers the compiler put an
  // s_waitcnt vmcnt(0) in front of the first fragment read behind the issue -- every K-step waited for its own fresh loads.)
  const unsigned apl_b = (unsigned)(p.dy_plane * 2), xpl_b = (unsigned)(p.x_plane * 2);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  // byte offset of a lane's 8-channel chunk = pixel row * mul + add.  Planar planes [row][ld]: mul = 2 ld, add = 2 channel.  BLOCKED planes
  // [channel / 16][row][16] -- what igemm_h2w_kernel reads, so that ONE plane set per tensor serves all three directions (round 5: the planar
  // set and its share of the split pass' write traffic are gone): mul = 32, add = (channel / 16) rows 32 + (channel & 8) 2.  A DMA instruction
  // (2 pixel rows x 32 chunks) then touches 16 pieces of 64 contiguous bytes instead of 2 of 512; the neighbouring waves fetch the other halves
  // of the same 128-byte lines in the same K-step.
  const unsigned amul = p.blocked ? 32u : (unsigned)p.ldo * 2u, bmul = p.blocked ? 32u : (unsigned)p.ldx * 2u;
  unsigned aadd[2], badd[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned ac = (unsigned)(acol[j] < 0 ? 0 : acol[j]), bc = (unsigned)bch[j];
    aadd[j] = p.blocked ? (ac >> 4) * (unsigned)p.P * 32u + (ac & 8u) * 2u : ac * 2u;
    badd[j] = p.blocked ? (bc >> 4) * (unsigned)p.x_rows * 32u + (bc & 8u) * 2u : bc * 2u;
  }
  unsigned VA[2], VB[2];
  auto prepA = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = (int)(arow[j] < r_end) & (int)(acol[j] >= 0);
      VA[j] = ok ? (unsigned)arow[j] * amul + aadd[j] : OOB;
      arow[j] += 16;
    }
  };
  auto prepB = [&](const int j) {
    const int iy = ty[j] * p.stride - p.pad + bky[j] * p.dil;
    const int ix = tx[j] * p.stride - p.pad + bkx[j] * p.dil;
    const bool ok = (int)(brow[j] < r_end) & (int)bcv[j] & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);
    VB[j] = ok ? (unsigned)((tb[j] * p.H + iy) * p.W + ix) * bmul + badd[j] : OOB;
    brow[j] += 16;                                   // advance this row by 16 pixels
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
  // The LDS-DMA is issued from inline asm: behind an LDS-DMA it can see (builtin), hipcc puts an s_waitcnt vmcnt(0) in front of the next
  // ds_read_b64_tr_b16 (it cannot prove that the transposed read does not alias the DMA's destination) -- every K-step then waited
  // for the loads it had just issued.  Hidden, the DMA is retired by the counted s_waitcnt at the top of the K loop alone (there is
  // no other vector-memory load in the loop for the compiler to count).  M0 = the wave's LDS destination; saved and restored.
  const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(smem) + (unsigned)wave * 1024u;
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  auto rsrc_words = [](const void* base, unsigned bytes) {    // raw buffer descriptor: base, stride 0, num_records = bytes, flags as above
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    return i32x4_{(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
  };
  const i32x4_ rA = rsrc_words(p.dy, 2u * apl_b), rB = rsrc_words(p.x, 2u * xpl_b);
  auto dma16 = [&](const i32x4_ rs, const unsigned voff, const unsigned soff, const unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rs), "s"(soff), "s"(dst)
                 : "memory");
  };
  auto piece = [&](const int buf, const int i) {     // i = 4 * plane + {A0, A1, B0, B1}
    const int pl = i >> 2, w = i & 3;
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * SLAB));
    if (w < 2) dma16(rA, VA[w], pl * apl_b, base + (unsigned)(pl * PLANE + w * 4096));
    else dma16(rB, VB[w - 2], pl * xpl_b, base + (unsigned)((2 + pl) * PLANE + (w - 2) * 4096));
  };
  auto issue = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(buf, i);
  };

  // fragment addresses (transposed reads): lane = 16 g + 4 q + pp inside its 32-lane half
  int ra[TM], rb[TN];
  {
    const int g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
#ifdef H2T_STAGE2
    const int rowb = (8 * h + q) * 512;               // pixel row 8h + q (+ 4 for the second read: + 2048 bytes)
    const int hm = wm * 256, hn = wn * 256;
#else
    const int rowb = h * 4096 + q * 256;              // pixel row 8h + q: row group 2h (+ 1 for the second read: + 2048 bytes), row q inside it
    const int hm = wm * 1024, hn = wn * 1024;         // the wave's half of the columns
#endif
#pragma unroll
    for (int t = 0; t < TM; ++t) ra[t] = rowb + hm + ((((t ^ q) << 2) + 2 * g1 + (pp >> 1)) << 4) + ((pp & 1) << 3);
#pragma unroll
    for (int u = 0; u < TN; ++u) rb[u] = 2 * PLANE + rowb + hn + ((((u ^ q) << 2) + 2 * g1 + (pp >> 1)) << 4) + ((pp & 1) << 3);
  }
  half8 Ah[TM], Bh[TN], Al[TM], Bl[TN];
#define T_READ(dst, base, off)                                                                              \
  do {                                                                                                      \
    const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((base) + (off)));                \
    const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((base) + (off) + 2048));         \
    typedef short s16x8_ __attribute__((ext_vector_type(8)));                                               \
    const s16x8_ v_ = {lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]};                       \
    dst = __builtin_bit_cast(half8, v_);                                                                    \
  } while (0)
#define H2_MFMA(A_, t_, B_, u_)                                                                              \
  do {                                                                                                        \
    acc[t_][u_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[t_], B_[u_], acc[t_][u_], 0, 0, 0);                \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)

  if (nks > 0) {
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {    // (past the end of the slab every offset is out of range = zeros)
      prep();
      issue(sl);
    }
    prep();
    __builtin_amdgcn_s_waitcnt(h2_waitcnt(8 * (NSLOT - 1), 0));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
#pragma unroll
      for (int t = 0; t < TM; ++t) { T_READ(Al[t], smem + ra[t], 1 * PLANE); T_READ(Bh[t], smem + rb[t], 0 * PLANE); }
#pragma unroll
      for (int t = 0; t < TM; ++t) { T_READ(Ah[t], smem + ra[t], 0 * PLANE); T_READ(Bl[t], smem + rb[t], 1 * PLANE); }
    }
    // lh(0)
    H2_MFMA(Al, 0, Bh, 0); H2_MFMA(Al, 0, Bh, 1); H2_MFMA(Al, 0, Bh, 2); H2_MFMA(Al, 0, Bh, 3);
    H2_MFMA(Al, 1, Bh, 0); H2_MFMA(Al, 1, Bh, 1); H2_MFMA(Al, 1, Bh, 2); H2_MFMA(Al, 1, Bh, 3);
    H2_MFMA(Al, 2, Bh, 0); H2_MFMA(Al, 2, Bh, 1); H2_MFMA(Al, 2, Bh, 2); H2_MFMA(Al, 2, Bh, 3);
    H2_MFMA(Al, 3, Bh, 0); H2_MFMA(Al, 3, Bh, 1); H2_MFMA(Al, 3, Bh, 2); H2_MFMA(Al, 3, Bh, 3);
    int nxt = 1, fill = 0;
    for (int k = 0; k < nks; ++k) {
      // all but my newest 8 (NSLOT - 2) LDS-DMA pieces have landed: step k+1 is there; every fragment read has returned; barrier:
      // ... everybody's, and every wave has read the fragments of step k out of its slot, which step k+NSLOT now overwrites
      __builtin_amdgcn_s_waitcnt(h2_waitcnt(8 * (NSLOT - 2), 0));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* aa_[TM]; const char* bb_[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) { aa_[t] = smem + nxt * SLAB + ra[t]; bb_[t] = smem + nxt * SLAB + rb[t]; }
      // hh(k); Al(k+1); the eight pieces of step k+NSLOT
      T_READ(Al[0], aa_[0], PLANE); H2_MFMA(Ah, 0, Bh, 0);
      T_READ(Al[1], aa_[1], PLANE); H2_MFMA(Ah, 0, Bh, 1);
      T_READ(Al[2], aa_[2], PLANE); H2_MFMA(Ah, 0, Bh, 2);
      T_READ(Al[3], aa_[3], PLANE); H2_MFMA(Ah, 0, Bh, 3);
      piece(fill, 0); H2_MFMA(Ah, 1, Bh, 0);
      piece(fill, 1); H2_MFMA(Ah, 1, Bh, 1);
      piece(fill, 2); H2_MFMA(Ah, 1, Bh, 2);
      piece(fill, 3); H2_MFMA(Ah, 1, Bh, 3);
      piece(fill, 4); H2_MFMA(Ah, 2, Bh, 0);
      piece(fill, 5); H2_MFMA(Ah, 2, Bh, 1);
      piece(fill, 6); H2_MFMA(Ah, 2, Bh, 2);
      piece(fill, 7); H2_MFMA(Ah, 2, Bh, 3);
      H2_MFMA(Ah, 3, Bh, 0); H2_MFMA(Ah, 3, Bh, 1); H2_MFMA(Ah, 3, Bh, 2); H2_MFMA(Ah, 3, Bh, 3);
      // hl(k); Bh(k+1); the addresses of step k+4
      T_READ(Bh[0], bb_[0], 0); H2_MFMA(Ah, 0, Bl, 0);
      T_READ(Bh[1], bb_[1], 0); H2_MFMA(Ah, 0, Bl, 1);
      T_READ(Bh[2], bb_[2], 0); H2_MFMA(Ah, 0, Bl, 2);
      T_READ(Bh[3], bb_[3], 0); H2_MFMA(Ah, 0, Bl, 3);
      H2_MFMA(Ah, 1, Bl, 0);
      prepA(); H2_MFMA(Ah, 1, Bl, 1);
      H2_MFMA(Ah, 1, Bl, 2); H2_MFMA(Ah, 1, Bl, 3);
      prepB(0); H2_MFMA(Ah, 2, Bl, 0);
      H2_MFMA(Ah, 2, Bl, 1); H2_MFMA(Ah, 2, Bl, 2);
      prepB(1); H2_MFMA(Ah, 2, Bl, 3);
      H2_MFMA(Ah, 3, Bl, 0); H2_MFMA(Ah, 3, Bl, 1); H2_MFMA(Ah, 3, Bl, 2); H2_MFMA(Ah, 3, Bl, 3);
      // lh(k+1) (step nks is all zeros); Ah(k+1), Bl(k+1)
      T_READ(Ah[0], aa_[0], 0); H2_MFMA(Al, 0, Bh, 0);
      T_READ(Ah[1], aa_[1], 0); H2_MFMA(Al, 0, Bh, 1);
      T_READ(Ah[2], aa_[2], 0); H2_MFMA(Al, 0, Bh, 2);
      T_READ(Ah[3], aa_[3], 0); H2_MFMA(Al, 0, Bh, 3);
      T_READ(Bl[0], bb_[0], PLANE); H2_MFMA(Al, 1, Bh, 0);
      T_READ(Bl[1], bb_[1], PLANE); H2_MFMA(Al, 1, Bh, 1);
      T_READ(Bl[2], bb_[2], PLANE); H2_MFMA(Al, 1, Bh, 2);
      T_READ(Bl[3], bb_[3], PLANE); H2_MFMA(Al, 1, Bh, 3);
      H2_MFMA(Al, 2, Bh, 0); H2_MFMA(Al, 2, Bh, 1); H2_MFMA(Al, 2, Bh, 2); H2_MFMA(Al, 2, Bh, 3);
      H2_MFMA(Al, 3, Bh, 0); H2_MFMA(Al, 3, Bh, 1); H2_MFMA(Al, 3, Bh, 2); H2_MFMA(Al, 3, Bh, 3);
      nxt = nxt == NSLOT - 1 ? 0 : nxt + 1;
      fill = fill == NSLOT - 1 ? 0 : fill + 1;
    }
  }
#undef T_READ
#undef H2_MFMA

  const int ea = -*p.edy, eb = -*p.ex;
  float* cout = p.C + (long long)split * p.c_split_stride;
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int col = n0 + wn * 128 + u * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < p.M && col < p.N) cout[(long long)row * p.ldc + col] = __builtin_ldexpf(__builtin_ldexpf(acc[t][u][r], ea), eb);
      }
    }
}

// out[i] = sum_s slab[s][i]  (fixed order: deterministic)
__global__ void h2_reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n4, int splits, long long stride4) {
  const f32x4* s = (const f32x4*)slabs;
  f32x4* o = (f32x4*)out;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 a = s[i];
    for (int k = 1; k < splits; ++k) a += s[i + k * stride4];
    o[i] = a;
  }
}

int h2t_splits(int tiles, long long P) {
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

int run_h2(H2Args a, hipStream_t st) {
  CS_REQUIRE(a.a_plane * 4 < (1ll << 32) - 64 && a.w_plane * 4 < (1ll << 32) - 64, "f16x2 blocked planes: an operand's two planes must stay below 4 GB");
  CS_REQUIRE(a.ea && a.ew, "f16x2: prescale exponents missing");
  a.tilesM = (a.M + 255) / 256;
  a.tilesN = ((a.zero_to > a.N ? a.zero_to : a.N) + 255) / 256;
  if (catseg_g_h2w_waves == 8) {
    // FAST epilogue: every row tile full, nothing but the previous contents added, no pad columns to zero (a ragged last column tile is fine)
    if (a.M % 256 == 0 && a.residual == nullptr && a.zero_to <= a.N && !catseg_g_h2w_slow_epilogue)
      hipLaunchKernelGGL(igemm_h2w8_kernel<true>, dim3(a.tilesM * a.tilesN), dim3(512), 0, st, a);
    else
      hipLaunchKernelGGL(igemm_h2w8_kernel<false>, dim3(a.tilesM * a.tilesN), dim3(512), 0, st, a);
  }
  else
    hipLaunchKernelGGL(igemm_h2w_kernel, dim3(a.tilesM * a.tilesN), dim3(256), 0, st, a);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

int amax_launch(const float* x, int ld, long long rows, int C, unsigned* amax_bits, hipStream_t st) {
  if (hipMemsetAsync(amax_bits, 0, 4, st) != hipSuccess) { catseg_set_error("amax: memset failed"); return CATSEG_EHIP; }
  const long long n = rows * ((C + 3) / 4);
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ld, rows, C, amax_bits);
  return CATSEG_OK;
}

// max|x| of a tensor whose producers left amax records (csrc/common.h: 16 slots each): the maximum over up to four records (the channel
// slices of a concatenation have one producer each) instead of a pass over the tensor
struct AmaxRecs { const unsigned* r[4]; };
__global__ void amax_merge_kernel(AmaxRecs recs, unsigned* __restrict__ out) {
  unsigned m = 0;
  const int i = threadIdx.x >> 4, s = threadIdx.x & 15;      // 64 threads: record i, slot s
  if (recs.r[i]) m = recs.r[i][s * CS_AMAX_STRIDE];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if (threadIdx.x == 0) *out = m;
}

}  // namespace

// scale (device, 8 bytes: {uint32 amax bits, int32 exponent}) is working storage + result of the split calls below
extern "C" size_t catseg_split2h_blocked_elems(long long rows, int C) { return (size_t)2 * (size_t)((C + 15) / 16) * 16 * (size_t)rows; }
extern "C" size_t catseg_split2h_planar_elems(long long rows, int C) { return (size_t)2 * (size_t)rows * ((C + 7) & ~7); }

// x [rows][ld] fp32 -> two fp16 planes of x * 2^e, e = 14 - floor(log2(max|x|)): blocked_planes [2][ceil(C/16)][rows][16] and / or
// planar_planes [2][rows][roundup(C, 8)] (either may be null); scale receives {bits of max|x|, e}.  Two launches (amax, split).
extern "C" int catseg_split2h(const float* x, long long rows, int C, int ld, void* blocked_planes, void* planar_planes, void* scale,
                              catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && ld >= C && ld % 4 == 0 && cs_aligned16(x) && cs_aligned16(blocked_planes) && cs_aligned16(planar_planes) &&
                 (blocked_planes || planar_planes) && scale && (((uintptr_t)scale) & 7) == 0,
             "split2h: bad args (ld must be a multiple of 4, pointers 16-byte aligned)");
  CS_REQUIRE((rows + 63) / 64 < (1ll << 31), "split2h: too many rows");
  hipStream_t st = (hipStream_t)stream;
  unsigned* ab = (unsigned*)scale;
  if (int rc = amax_launch(x, ld, rows, C, ab, st)) return rc;
  const int c16 = (C + 15) / 16, ldp = (C + 7) & ~7;
  hipLaunchKernelGGL(split2h_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)((c16 * 16 + 127) / 128)), dim3(256), 0, st, x, ld, rows, C, ldp,
                     (const unsigned*)ab, (u16*)blocked_planes, (long long)c16 * rows * 16, (u16*)planar_planes, rows * ldp, (int*)(ab + 1));
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_split2h without its pass over x for max|x|: the producers of x left amax records (catseg_bn_apply_amax, catseg_bn_backward_amax,
// catseg_add_n_act_amax ...; bilinear resizing and copies keep their input's), up to four of them for the channel slices of a concatenation
// (null = unused).  The records must bound |x| (an exact maximum or an upper bound: the exponent only has to keep x * 2^e inside fp16).
extern "C" int catseg_split2h_bound(const float* x, long long rows, int C, int ld, void* blocked_planes, void* planar_planes, void* scale,
                                    const void* rec0, const void* rec1, const void* rec2, const void* rec3, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && ld >= C && ld % 4 == 0 && cs_aligned16(x) && cs_aligned16(blocked_planes) && cs_aligned16(planar_planes) &&
                 (blocked_planes || planar_planes) && scale && (((uintptr_t)scale) & 7) == 0,
             "split2h_bound: bad args (ld must be a multiple of 4, pointers 16-byte aligned)");
  CS_REQUIRE(rec0 || rec1 || rec2 || rec3, "split2h_bound: no amax record");
  CS_REQUIRE((rows + 63) / 64 < (1ll << 31), "split2h_bound: too many rows");
  hipStream_t st = (hipStream_t)stream;
  unsigned* ab = (unsigned*)scale;
  AmaxRecs recs = {{(const unsigned*)rec0, (const unsigned*)rec1, (const unsigned*)rec2, (const unsigned*)rec3}};
  hipLaunchKernelGGL(amax_merge_kernel, dim3(1), dim3(64), 0, st, recs, ab);
  const int c16 = (C + 15) / 16, ldp = (C + 7) & ~7;
  hipLaunchKernelGGL(split2h_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)((c16 * 16 + 127) / 128)), dim3(256), 0, st, x, ld, rows, C, ldp,
                     (const unsigned*)ab, (u16*)blocked_planes, (long long)c16 * rows * 16, (u16*)planar_planes, rows * ldp, (int*)(ab + 1));
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// blocked planes [2][sum C_i / 16][B Ho Wo][16] of cat_i(bilinear(x_i -> Ho x Wo, align_corners = False)) for up to four NHWC sources
// x_i [B][H_i][W_i][C_i] (row stride ld_i, C_i % 16 == 0; a source of the output's size is copied), without the fp32 concatenation; scale as
// catseg_split2h_bound: {bits of max over the sources' amax records, e}.
extern "C" int catseg_concat_bilinear_split2h(int nsrc, const float* const* xs, const int* lds, const int* Hs, const int* Ws, const int* Cs,
                                              const void* const* records, int B, int Ho, int Wo, void* blocked_planes, void* scale,
                                              catseg_stream_t stream) {
  CS_REQUIRE(nsrc >= 1 && nsrc <= 4 && xs && lds && Hs && Ws && Cs && records && B > 0 && Ho > 0 && Wo > 0 && blocked_planes && scale &&
                 cs_aligned16(blocked_planes) && (((uintptr_t)scale) & 7) == 0, "concat_bilinear_split2h: bad args");
  CatArgs a = {};
  AmaxRecs recs = {{nullptr, nullptr, nullptr, nullptr}};
  int chunk = 0;
  for (int i = 0; i < nsrc; ++i) {
    CS_REQUIRE(xs[i] && records[i] && Cs[i] > 0 && Cs[i] % 16 == 0 && lds[i] >= Cs[i] && lds[i] % 4 == 0 && cs_aligned16(xs[i]) && Hs[i] > 0 && Ws[i] > 0,
               "concat_bilinear_split2h: source %d (channels must be a multiple of 16, ld of 4, an amax record per source)", i);
    a.s[i].x = xs[i]; a.s[i].ld = lds[i]; a.s[i].H = Hs[i]; a.s[i].W = Ws[i]; a.s[i].chunk0 = chunk;
    a.s[i].sh = (float)Hs[i] / (float)Ho; a.s[i].sw = (float)Ws[i] / (float)Wo;
    recs.r[i] = (const unsigned*)records[i];
    chunk += Cs[i] / 16;
  }
  a.nsrc = nsrc; a.nchunks = chunk; a.B = B; a.Ho = Ho; a.Wo = Wo; a.rows = (long long)B * Ho * Wo;
  CS_REQUIRE(a.rows * chunk * 64 < (1ll << 32) - 64, "concat_bilinear_split2h: the two planes must stay below 4 GB");
  hipStream_t st = (hipStream_t)stream;
  unsigned* ab = (unsigned*)scale;
  hipLaunchKernelGGL(amax_merge_kernel, dim3(1), dim3(64), 0, st, recs, ab);
  hipLaunchKernelGGL(concat_bilinear_split2h_kernel, dim3((unsigned)((a.rows + 255) / 256), (unsigned)chunk), dim3(256), 0, st, a, (const unsigned*)ab,
                     (u16*)blocked_planes, (long long)chunk * a.rows * 16, (int*)(ab + 1));
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// OHWI weights viewed as [Cout][K = taps * Cin] -> blocked fp16 planes [2][K/16][Cout][16] (Cin % 16 == 0): B operand of the forward conv
extern "C" int catseg_split2h_weight_blocked(const float* w, int O, int taps, int Cin, void* planes, void* scale, catseg_stream_t stream) {
  CS_REQUIRE(O > 0 && taps > 0 && Cin > 0 && Cin % 16 == 0 && cs_aligned16(planes) && cs_aligned16(w) && scale, "split2h_weight_blocked: needs Cin % 16 == 0");
  hipStream_t st = (hipStream_t)stream;
  unsigned* ab = (unsigned*)scale;
  const int K = taps * Cin;
  if (int rc = amax_launch(w, K, O, K, ab, st)) return rc;
  const long long n = (long long)O * K;
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split2h_weight_blocked_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, w, O, K, O, O, taps, Cin, (const unsigned*)ab,
                     (u16*)planes, n, (int*)(ab + 1));
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// OHWI weights -> blocked fp16 planes of the transposed filter bank [2][taps * Opad / 16][Cin][16], Opad = roundup(O, 16): B operand of
// backward-data
extern "C" int catseg_split2h_weight_t_blocked(const float* w, int O, int taps, int Cin, void* planes, void* scale, catseg_stream_t stream) {
  CS_REQUIRE(O > 0 && taps > 0 && Cin > 0 && Cin % 4 == 0 && cs_aligned16(planes) && cs_aligned16(w) && scale, "split2h_weight_t_blocked: bad args");
  hipStream_t st = (hipStream_t)stream;
  unsigned* ab = (unsigned*)scale;
  if (int rc = amax_launch(w, taps * Cin, O, taps * Cin, ab, st)) return rc;
  const int Opad = (O + 15) & ~15;
  const int K = taps * Opad;
  const long long n = (long long)Cin * K;
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split2h_weight_blocked_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, w, Cin, K, O, Opad, taps, Cin, (const unsigned*)ab,
                     (u16*)planes, n, (int*)(ab + 1));
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

namespace {
H2Args h2_fwd_args(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* w_planes, const void* w_scale,
                   const float* bias, float* y) {
  H2Args a = {};
  a.a_rows = d->B * d->H * d->W; a.w_rows = d->Cout;
  a.a = (const u16*)x_planes; a.a_plane = (long long)a.a_rows * d->Cin;
  a.w = (const u16*)w_planes; a.w_plane = (long long)d->Cout * d->kh * d->kw * d->Cin;
  a.ea = (const int*)x_scale + 1; a.ew = (const int*)w_scale + 1;
  a.C = y; a.ldc = d->ldy; a.bias = bias;
  a.M = d->B * d->Ho * d->Wo; a.N = d->Cout; a.Cin = d->Cin; a.taps = d->kh * d->kw;
  a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.sign = 1;
  return a;
}
}  // namespace

// y = conv(x, w) (+ bias) from two-plane fp16 operands: x_planes / x_scale = the blocked planes of catseg_split2h of x (C = Cin, Cin % 16 == 0),
// w_planes / w_scale = catseg_split2h_weight_blocked.  bn_part may be null (then tile_rows / n_tiles are not touched).
extern "C" int catseg_conv2d_fwd_f16x2_blocked(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* w_planes,
                                               const void* w_scale, const float* bias, float* y, int zero_to, float* bn_part,
                                               size_t bn_part_floats, int* tile_rows, int* n_tiles, catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->Cin % 16 == 0 && d->kh * d->kw <= 32, "conv fwd f16x2: needs Cin % 16 == 0, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(w_planes) && cs_aligned16(y) && zero_to <= d->ldy && x_scale && w_scale, "conv fwd f16x2: alignment");
  H2Args a = h2_fwd_args(d, x_planes, x_scale, w_planes, w_scale, bias, y);
  a.zero_to = zero_to;
  if (bn_part != nullptr) {
    CS_REQUIRE(tile_rows && n_tiles, "conv fwd f16x2: tile_rows / n_tiles");
    const int nt = (a.M + 255) / 256;
    *tile_rows = 0; *n_tiles = 0;
    if ((size_t)nt * 3 * d->Cout <= bn_part_floats) {
      a.bn_part = bn_part;
      *tile_rows = 256; *n_tiles = nt;
    }
  }
  return run_h2(a, (hipStream_t)stream);
}

// inference: y = act(conv(x, w) + bias (+ residual)) in one kernel (the counterpart of catseg_conv2d_fwd_fused_bf16x3_blocked)
extern "C" int catseg_conv2d_fwd_fused_f16x2_blocked(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* w_planes,
                                                     const void* w_scale, const float* bias, const float* residual, int ldr, int relu, float* y,
                                                     catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->Cin % 16 == 0 && d->kh * d->kw <= 32, "conv fwd fused f16x2: needs Cin % 16 == 0, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(w_planes) && cs_aligned16(y) && (residual == nullptr || ldr >= d->Cout) && x_scale && w_scale,
             "conv fwd fused f16x2: bad args");
  H2Args a = h2_fwd_args(d, x_planes, x_scale, w_planes, w_scale, bias, y);
  a.residual = residual; a.ldr = ldr; a.relu = relu;
  return run_h2(a, (hipStream_t)stream);
}

// dx (+)= conv_transpose(dy, w), stride 1: dy_planes / dy_scale = the blocked planes of catseg_split2h of dy (C = Cout; the channel tail up to
// roundup(Cout, 16) is zero), wt_planes / wt_scale = catseg_split2h_weight_t_blocked
extern "C" int catseg_conv2d_bwd_data_f16x2_blocked(const catseg_conv_desc* d, const void* dy_planes, const void* dy_scale, const void* wt_planes,
                                                    const void* wt_scale, float* dx, int accumulate, catseg_stream_t stream) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->stride == 1 && d->kh * d->kw <= 32, "conv bwd_data f16x2: stride 1, <= 32 taps, dense");
  CS_REQUIRE(cs_aligned16(dy_planes) && cs_aligned16(wt_planes) && cs_aligned16(dx) && dy_scale && wt_scale, "conv bwd_data f16x2: alignment");
  const int cop = (d->Cout + 15) & ~15;
  H2Args a = {};
  a.a_rows = d->B * d->Ho * d->Wo; a.w_rows = d->Cin;
  a.a = (const u16*)dy_planes; a.a_plane = (long long)a.a_rows * cop;
  a.w = (const u16*)wt_planes; a.w_plane = (long long)d->Cin * d->kh * d->kw * cop;
  a.ea = (const int*)dy_scale + 1; a.ew = (const int*)wt_scale + 1;
  a.C = dx; a.ldc = d->ldx; a.bias = nullptr;
  a.M = d->B * d->H * d->W; a.N = d->Cin; a.Cin = cop; a.taps = d->kh * d->kw;
  a.H = d->Ho; a.W = d->Wo; a.Ho = d->H; a.Wo = d->W; a.kw = d->kw; a.stride = 1; a.pad = d->pad; a.dil = d->dil;
  a.sign = -1; a.accumulate = accumulate;
  return run_h2(a, (hipStream_t)stream);
}

// workspace: split-reduction slabs (only when more than one split is planned)
extern "C" size_t catseg_conv2d_bwd_weight_f16x2_workspace(const catseg_conv_desc* d) {
  if (!d) return 0;
  const long long P = (long long)d->B * d->Ho * d->Wo;
  const int N = d->kh * d->kw * d->Cin;
  const int tiles = ((d->Cout + 255) / 256) * ((N + 255) / 256);
  const int sp = h2t_splits(tiles, P);
  return sp > 1 ? cs_align_up((size_t)sp * d->Cout * N * 4, 256) : 0;
}

// dw[o][ky][kx][c] = sum_p dy[p][o] x[pix(p,ky,kx)][c] from two-plane fp16 operands in the PLANAR layout of catseg_split2h:
// dy_planes / dy_scale of dy (C = Cout), x_planes / x_scale of x (C = Cin, Cin % 8 == 0)
namespace {
int run_h2t(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* dy_planes, const void* dy_scale, float* dw,
            void* workspace, size_t workspace_bytes, int blocked, hipStream_t st) {
  CS_REQUIRE(d && !d->stem4 && d->groups <= 1 && d->Cin % 8 == 0, "conv bwd_weight f16x2: dense, Cin % 8 == 0");
  CS_REQUIRE(cs_aligned16(x_planes) && cs_aligned16(dy_planes) && cs_aligned16(dw) && x_scale && dy_scale, "conv bwd_weight f16x2: alignment");
  const int cpad_x = blocked ? (d->Cin + 15) & ~15 : d->Cin, cpad_o = blocked ? (d->Cout + 15) & ~15 : (d->Cout + 7) & ~7;
  CS_REQUIRE((long long)d->B * d->H * d->W * cpad_x * 4 < (1ll << 32) - 64 && (long long)d->B * d->Ho * d->Wo * cpad_o * 4 < (1ll << 32) - 64,
             "conv bwd_weight f16x2: an operand's two planes must stay below 4 GB");
  const size_t need = catseg_conv2d_bwd_weight_f16x2_workspace(d);
  if (workspace_bytes < need || (need && !workspace)) {
    catseg_set_error("conv bwd_weight f16x2: workspace %zu < %zu", workspace_bytes, need);
    return CATSEG_EWORKSPACE;
  }
  H2TArgs a = {};
  a.P = d->B * d->Ho * d->Wo;
  a.M = d->Cout; a.Cin = d->Cin; a.taps = d->kh * d->kw; a.N = a.taps * d->Cin;
  a.ldo = (d->Cout + 7) & ~7; a.dy = (const u16*)dy_planes; a.dy_plane = (long long)a.P * cpad_o;
  a.ldx = d->Cin; a.x = (const u16*)x_planes; a.x_rows = (long long)d->B * d->H * d->W; a.x_plane = a.x_rows * cpad_x;
  a.blocked = blocked;
  a.edy = (const int*)dy_scale + 1; a.ex = (const int*)x_scale + 1;
  a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  const int img = d->Ho * d->Wo;
  a.step_b = 16 / img; a.step_qy = (16 % img) / d->Wo; a.step_rx = (16 % img) % d->Wo;
  a.tilesM = (a.M + 255) / 256; a.tilesN = (a.N + 255) / 256;
  const int sp0 = h2t_splits(a.tilesM * a.tilesN, a.P);
  a.rows_per_split = (int)((((long long)a.P + sp0 - 1) / sp0 + 15) / 16 * 16);
  const int sp = (a.P + a.rows_per_split - 1) / a.rows_per_split;
  a.ldc = a.N; a.c_split_stride = (long long)a.M * a.N;
  a.C = sp > 1 ? (float*)workspace : dw;
  hipLaunchKernelGGL(igemm_h2t_kernel, dim3(a.tilesM * a.tilesN * sp), dim3(256), 0, st, a);
  CS_LAUNCH_CHECK();
  if (sp > 1) {
    const long long n4 = (long long)a.M * a.N / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(h2_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n4, sp, n4);
    CS_LAUNCH_CHECK();
  }
  return CATSEG_OK;
}
}  // namespace

extern "C" int catseg_conv2d_bwd_weight_f16x2(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* dy_planes,
                                              const void* dy_scale, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  return run_h2t(d, x_planes, x_scale, dy_planes, dy_scale, dw, workspace, workspace_bytes, 0, (hipStream_t)stream);
}

// the same from the BLOCKED planes of catseg_split2h ([2][ceil(C / 16)][rows][16], channel tails zero) -- the planes the forward and the
// backward-data kernel read: one plane set per tensor serves all three directions
extern "C" int catseg_conv2d_bwd_weight_f16x2_blocked(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* dy_planes,
                                                      const void* dy_scale, float* dw, void* workspace, size_t workspace_bytes,
                                                      catseg_stream_t stream) {
  return run_h2t(d, x_planes, x_scale, dy_planes, dy_scale, dw, workspace, workspace_bytes, 1, (hipStream_t)stream);
}
