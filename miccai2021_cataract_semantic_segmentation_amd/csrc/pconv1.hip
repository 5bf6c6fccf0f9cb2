// Pointwise (1 x 1, stride 1) convolution of the layers that are NOT large enough for the blocked-plane kernel of igemm_f16x2.hip and were
// the bulk of the fp32-MFMA residue of an OCRNet-HRNet-W48 step: the stage-1 bottlenecks (models/HRNetv2.py:68-106 of the reference: 1 x 1
// 64 -> 64 / 64 -> 256 / 256 -> 64 on 261 120 pixels), the object-attention block of the OCR head (models/OCR.py:186-235: f_pixel 512 -> 256
// -> 256, f_up 256 -> 512) and the 1 x 1 fuse layers of the HRNet modules (models/HRNetv2.py:237-261).  With K = Cin of 64 ... 512 these
// GEMMs are HBM-bound (25 ... 85 FLOP per byte), and a separate split pass over the activation costs what the fp16 matrix cores gain
// (tools/ab_1x1_hr.py, round 4).  Here the split happens IN REGISTERS:
//
//   * arithmetic of igemm_f16x2.hip: xs = x * 2^e (e from the tensor's amax record, left by its producer),  h = fp16(xs),  l = fp16(xs - h),
//     a.b ~ hh + hl + lh into one fp32 accumulator (three v_mfma_f32_32x32x16_f16), result scaled back by 2^-(e_x + e_w);
//   * A operand (pixel rows): a lane of the 32 x 32 x 16 MFMA owns 8 consecutive k of one row = 32 contiguous bytes of the fp32 NHWC row --
//     it loads them straight from global memory (two 16-byte buffer loads, rows past M and channels past K come back as zeros), scales,
//     splits and holds the two fragments: no LDS, no barrier on the activation's path, 4 bytes read per element, nothing written;
//   * B operand (weights): a pre-split image per layer and direction (p1_prep_batch_kernel: all layers of a network in one launch per
//     step) streams through three LDS slots by LDS-DMA, one 32-deep k-group per slot, shared by the block's eight waves;
//   * block = 256 rows x up to 256 columns, 8 waves of 32 rows (two per SIMD, <= 256 registers), one barrier per k-group;
//   * epilogue: bias, accumulate, BatchNorm partials per (row tile, channel) (common.h: cs_tile_bn_partials) as the other forward kernels.
//
// Backward-data of the same layers is the same kernel on dy with the transposed weight image.  Backward-weight: p1t_kernel below.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// prescale exponent from the bits of a tensor's max |x| (igemm_f16x2.hip: h2_exponent): amax * 2^e in [2^14, 2^15)
__host__ __device__ inline int p1_exponent(unsigned amax_bits) {
  const int ex = (int)((amax_bits >> 23) & 0xFF);
  if (ex == 0 || ex == 255) return 0;
  const int e = 14 - (ex - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

__host__ __device__ constexpr int p1_bn(int N) { return N >= 256 ? 256 : (N + 31) / 32 * 32; }   // columns of a block tile
__host__ __device__ constexpr int p1_ntn(int N) { return (N + p1_bn(N) - 1) / p1_bn(N); }
__host__ __device__ constexpr int p1_groups(int K) { return ((K + 31) / 32 + 1) / 2 * 2; }        // 32-deep k-groups, an even count (the K loop is unrolled twice)

// ---- weight image: [k-group][n tile][k-tile 0/1][plane h/l][BN][16 halves], the two 8-half chunks of a row swapped where (n >> 3) & 1
// (conflict-free ds_read_b128 of the B fragments), zero beyond N / K.  transposed: B[n][k] = w[k][n] (backward-data: n = input channel)
// General form (round 5, the "gather" launches below): the weights are OHWI [O][kh][kw][I]; an image holds a SUB-GRID of the taps --
// ky = ky0 + a kys (a < nky), kx = kx0 + b kxs (b < nkx) -- with k' = (a nkx + b) Ck + c: forward Ck = I, c = input channel, n = o;
// transposed (backward-data) Ck = O, c = output channel, n = input channel.  1 x 1: kh = kw = nky = nkx = 1.
struct P1Entry { long long w_off, img_off; int O, I, kh, kw, transposed, ky0, kys, nky, kx0, kxs, nkx, pad; };

__global__ __launch_bounds__(256) void p1_amax_batch_kernel(const float* __restrict__ flat, const P1Entry* __restrict__ ent, unsigned* __restrict__ recs) {
  const P1Entry e = ent[blockIdx.y];
  const float* w = flat + e.w_off;
  const long long n = (long long)e.O * e.I * e.kh * e.kw;
  unsigned m = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m = max(m, __float_as_uint(w[i]) & 0x7FFFFFFFu);
  cs_amax_commit1(m, recs + 2 * blockIdx.y);
}

__global__ __launch_bounds__(256) void p1_prep_batch_kernel(const float* __restrict__ flat, const P1Entry* __restrict__ ent,
                                                            unsigned char* __restrict__ img_base, unsigned* __restrict__ recs) {
  const P1Entry e = ent[blockIdx.y];
  const int ex = p1_exponent(recs[2 * blockIdx.y]);
  if (blockIdx.x == 0 && threadIdx.x == 0) ((int*)recs)[2 * blockIdx.y + 1] = ex;
  const float* w = flat + e.w_off;
  const int N = e.transposed ? e.I : e.O, Ck = e.transposed ? e.O : e.I, K = e.nky * e.nkx * Ck;
  const int BN = p1_bn(N), ntn = p1_ntn(N), G = p1_groups(K);
  half8* img = (half8*)(img_base + e.img_off);
  const long long pieces = (long long)G * ntn * 2 * BN * 2;          // (g, nt, kt, nn, chunk): one thread writes the h and the l piece
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pieces; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i & 1);
    long long r = i >> 1;
    const int nn = (int)(r % BN); r /= BN;
    const int kt = (int)(r & 1); r >>= 1;
    const int nt = (int)(r % ntn);
    const int g = (int)(r / ntn);
    const int n = nt * BN + nn, k0 = 32 * g + 16 * kt + 8 * c;
    half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      float v = 0.f;
      if (n < N && k < K) {
        const int t = k / Ck, c = k - t * Ck;
        const int ta = t / e.nkx, tb = t - ta * e.nkx;
        const int ky = e.ky0 + ta * e.kys, kx = e.kx0 + tb * e.kxs;
        const int o = e.transposed ? c : n, ci = e.transposed ? n : c;
        v = w[(((long long)o * e.kh + ky) * e.kw + kx) * e.I + ci];
      }
      const float xs = __builtin_ldexpf(v, ex);
      const _Float16 hh = (_Float16)xs;
      h[j] = hh;
      l[j] = (_Float16)(xs - (float)hh);
    }
    const int pc = c ^ ((nn >> 3) & 1);
    const long long row0 = (((long long)(g * ntn + nt) * 2 + kt) * 2) * BN;      // plane 0 rows of this (g, nt, kt)
    img[(row0 + nn) * 2 + pc] = h;
    img[(row0 + BN + nn) * 2 + pc] = l;
  }
}

// how a GEMM row and a k' index of a gather launch map to an element of the A tensor, and a row to an output pixel
struct P1Geo {
  int Wg, HWg;              // row grid: row -> (image, i, j) = (row / HWg, (row % HWg) / Wg, row % Wg)
  int si, y0, x0;           // A pixel of tap (0, 0) at grid point (i, j): (i si + y0, j si + x0)
  int ya, xa, nb;           // tap t = (a, b) = (t / nb, t % nb): A pixel += (a ya, b xa)
  int Hi, Wi, C;            // A: images of Hi x Wi pixels with C channels (k' = t C + c)
  unsigned mC, mnb;         // ceil(2^32 / C), ceil(2^32 / nb) (nb > 1)
  int so, oy, ox, Ho, Wo;   // output pixel of grid point (i, j): (i so + oy, j so + ox) of an Ho x Wo image
  int linear_out;           // the row grid IS the output image: output row = GEMM row
};

struct P1Args {
  const float* x; int ldx;            // A: [M][ldx] fp32, columns [0, K)
  const unsigned* xrec;               // amax record of A (CS_AMAX_SLOTS slots)
  const unsigned char* wimg;          // weight image of (N, K)
  const int* wrec;                    // {amax bits, exponent} of the weights
  float* y; int ldy;
  const float* bias;
  int M, N, K;
  int tilesM, ntn;
  int accumulate;
  float* bn_part;                     // [tilesM][3][N] or null
  unsigned long long x_bytes, img_bytes;
  unsigned long long y_bytes;         // gather launches with gathered OUTPUT rows: bytes of the whole output (0: unknown -> general epilogue)
  P1Geo geo;                          // (GEO launches)
};

// cs_tile_bn_partials (common.h) for a FULL 256-row tile of p1_kernel -- 8 waves stacked along M, 16 accumulator rows per lane, lanes l and
// l ^ 32 hold the other rows of the same columns -- with the same operations in the same order (bit-identical partials) and a scheduling fence
// behind every column tile: in straight-line code the scheduler otherwise interleaves the eight column tiles' sums, keeps their temporaries alive
// together and spills ~300 registers per thread in the 256-column form.
template <int NS>
__device__ __forceinline__ void p1_full_tile_bn_partials(float* lds, int tile_cols, int l31, bool writer_lane, int wave_m, const f32x16 (&acc)[NS],
                                                         float* part, int C, int gcol0) {
  constexpr int WM = 8;
  float mean[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += acc[j][i];
    a += __shfl_xor(a, 32, 64);
    if (writer_lane) lds[wave_m * tile_cols + 32 * j + l31] = a;
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  const float inv = 1.f / 256.f;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) t += lds[w * tile_cols + 32 * j + l31];
    mean[j] = t * inv;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float d = acc[j][i] - mean[j];
      a += d;
      b += d * d;
    }
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if (writer_lane) {
      lds[wave_m * tile_cols + 32 * j + l31] = a;
      lds[(WM + wave_m) * tile_cols + 32 * j + l31] = b;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  if (writer_lane && wave_m == 0) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int gc = gcol0 + 32 * j + l31;
      if (gc < C) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          a += lds[w * tile_cols + 32 * j + l31];
          b += lds[(WM + w) * tile_cols + 32 * j + l31];
        }
        part[gc] = mean[j];
        part[C + gc] = a;
        part[2 * C + gc] = b;
      }
    }
  }
}

constexpr int p1_waitcnt(int vm) { return (vm & 15) | (7 << 4) | (15 << 8) | ((vm >> 4) << 14); }     // vmcnt only (expcnt / lgkmcnt: no wait)

template <int NT, bool GEO, bool FULL>
__device__ __forceinline__ void p1_body(const P1Args& p) {
  constexpr int BN = 32 * NT;
  constexpr int GB = 4 * BN * 32;                   // bytes of one k-group of B: k-tile (2) x plane (2) x BN rows x 32 bytes
  constexpr int PIECES = GB / 16;                   // = 256 NT: a multiple of 64
  constexpr int PB = (PIECES + 511) / 512;          // LDS-DMA instructions per wave per k-group
  constexpr int NBUF = 3;
  constexpr int EPI = 2 * 8 * BN * 4;               // floats the BatchNorm-partials exchange needs, in bytes
  constexpr int SMEM = (NBUF * GB > EPI ? NBUF * GB : EPI) + 1024;   // + a 1 KB sink for the idle waves' (zero-filled) trailing pieces
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;

  // block -> (row tile, column tile): the column tiles of one row tile sit 8 block ids apart, i.e. on the SAME XCD (blocks are dealt
  // round-robin over the eight XCDs): the second read of the activation rows hits that XCD's L2
  int tile_m = blockIdx.x, tile_n = 0;
  if (p.ntn > 1) {
    const int per = 8 * p.ntn, grp = blockIdx.x / per, r = blockIdx.x - grp * per;
    tile_n = r >> 3;
    tile_m = grp * 8 + (r & 7);
  }
  if (tile_m >= p.tilesM) return;
  const int m0 = tile_m * 256, n0 = tile_n * BN;

  const int ex = p1_exponent(cs_amax_read(p.xrec));
  const int G = p1_groups(p.K);

  f32x16 acc[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

  // A: a raw buffer over the activation (rows past M are past its end: zeros); a lane's k-group = four 16-byte loads
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)(unsigned)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.wimg, (short)0, (int)(unsigned)p.img_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int row = m0 + wave * 32 + l31;
  const unsigned klim = row < p.M ? (unsigned)p.K : 0u;          // rows past M: every k is "past K" (branch-free: the loads are issued
  unsigned a_row = 0;                                            //  unconditionally, the vmcnt bookkeeping below counts on it)
  int ybase = 0, xbase = 0;
  if (GEO) {
    const int rr = row < p.M ? row : 0;
    const int gb = rr / p.geo.HWg, rem = rr - gb * p.geo.HWg;
    const int gi = rem / p.geo.Wg, gj = rem - gi * p.geo.Wg;
    ybase = gi * p.geo.si + p.geo.y0;
    xbase = gj * p.geo.si + p.geo.x0;
    a_row = (unsigned)(gb * p.geo.Hi * p.geo.Wi);                // (pixel index of the image's origin)
  } else {
    a_row = (unsigned)row * (unsigned)p.ldx * 4u + (unsigned)hh * 32u;
  }
  auto a_off = [&](int g, int kt, int c) -> unsigned {
    const unsigned k8 = (unsigned)(32 * g + 16 * kt + 8 * hh);
    if (GEO) {
      const unsigned tap = __umulhi(k8, p.geo.mC), cc = k8 - tap * (unsigned)p.geo.C;
      const unsigned ta = p.geo.nb == 1 ? tap : __umulhi(tap, p.geo.mnb), tb = tap - ta * (unsigned)p.geo.nb;
      const int yi = ybase + (int)ta * p.geo.ya, xi = xbase + (int)tb * p.geo.xa;
      const bool ok = (int)(k8 + 4u * c < klim) & (int)((unsigned)yi < (unsigned)p.geo.Hi) & (int)((unsigned)xi < (unsigned)p.geo.Wi);
      const unsigned m = (unsigned)-(int)ok;
      const unsigned off = ((a_row + (unsigned)(yi * p.geo.Wi + xi)) * (unsigned)p.ldx + cc) * 4u + 16u * c;
      return (off & m) | (OOB & ~m);
    }
    const unsigned m = (unsigned)-(int)(k8 + 4u * c < klim);
    return ((a_row + (unsigned)(32 * g + 16 * kt) * 4u + 16u * c) & m) | (OOB & ~m);
  };
  // B: k-group g of column tile tile_n is one contiguous run of GB bytes of the image
  const unsigned b_lane = (unsigned)(wave * 64 + lane) * 16u;
  auto issueB = [&](int g, int buf) {
    const unsigned src0 = (unsigned)((g * p.ntn + tile_n)) * (unsigned)GB;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const bool live = i * 512 + wave * 64 < PIECES;              // (wave-uniform)
      const unsigned voff = live ? src0 + (unsigned)i * 8192u + b_lane : OOB;
      char* dst = live ? smem + buf * GB + (i * 512 + wave * 64) * 16 : smem + SMEM - 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, 0);
    }
  };

  f32x4 ra[2][4];       // raw fp32 of the two k-groups in flight: [parity][kt * 2 + c]
  auto loadA = [&](int g, int par) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[par][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off(g, q >> 1, q & 1), 0, 0));
  };
  half8 Ah[2], Al[2];
  auto splitA = [&](int par) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const f32x4 v0 = ra[par][kt * 2], v1 = ra[par][kt * 2 + 1];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xs = __builtin_ldexpf(j < 4 ? v0[j & 3] : v1[j & 3], ex);
        const _Float16 h = (_Float16)xs;
        Ah[kt][j] = h;
        Al[kt][j] = (_Float16)(xs - (float)h);
      }
    }
  };
  const int boff = l31 * 32 + ((hh ^ ((l31 >> 3) & 1)) << 4);
  // B fragments of (k-tile, column pair) q = 0 .. 2 NP - 1 are read one pair AHEAD of the MFMAs that use them
  constexpr int NP = (NT + 1) / 2;                   // column pairs per k-tile
  auto compute = [&](int buf) {
    const char* bb = smem + buf * GB + boff;
    half8 bh[2][2], bl[2][2];
    auto rd = [&](int q, int st) {
      const int kt = q / NP, u = 2 * (q - kt * NP);
      bh[st][0] = *(const half8*)(bb + ((kt * 2 + 0) * BN + 32 * u) * 32);
      bl[st][0] = *(const half8*)(bb + ((kt * 2 + 1) * BN + 32 * u) * 32);
      if (u + 1 < NT) {
        bh[st][1] = *(const half8*)(bb + ((kt * 2 + 0) * BN + 32 * (u + 1)) * 32);
        bl[st][1] = *(const half8*)(bb + ((kt * 2 + 1) * BN + 32 * (u + 1)) * 32);
      }
    };
    rd(0, 0);
#pragma unroll
    for (int q = 0; q < 2 * NP; ++q) {
      const int st = q & 1, kt = q / NP, u = 2 * (q - kt * NP);
      if (q + 1 < 2 * NP) rd(q + 1, st ^ 1);
      if (u + 1 < NT) {
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[kt], bh[st][0], acc[u], 0, 0, 0);
        acc[u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[kt], bh[st][1], acc[u + 1], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kt], bh[st][0], acc[u], 0, 0, 0);
        acc[u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kt], bh[st][1], acc[u + 1], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kt], bl[st][0], acc[u], 0, 0, 0);
        acc[u + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kt], bl[st][1], acc[u + 1], 0, 0, 0);
      } else {
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[kt], bh[st][0], acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kt], bh[st][0], acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[kt], bl[st][0], acc[u], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);             // (keeps the next pair's reads in front of this pair's MFMAs, not behind them)
    }
  };

  // Order of the vector-memory operations (vmcnt retires them in order): A(0) B(0) A(1) B(1) | A(2) [wait] B(2) | A(3) [wait] B(3) | ...
  // At the wait of iteration g everything but the newest A(g+1) B(g+1) A(g+2) may be outstanding: 8 + PB operations.
  loadA(0, 0);
  issueB(0, 0);
  loadA(1, 1);
  issueB(1, 1);
  for (int g = 0; g < G; g += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int gg = g + par;
      splitA(par);                                   // (the compiler waits for A(gg) here)
      loadA(gg + 2, par);                            // past the last group: out of range = zeros, never used
      __builtin_amdgcn_sched_barrier(0);             // (the count below assumes the four loads above have been issued)
      __builtin_amdgcn_s_waitcnt(p1_waitcnt(8 + PB));
      __builtin_amdgcn_s_barrier();                  // B(gg) of every wave has landed; every wave is done with slot (gg + 2) % 3 (step gg - 1)
      asm volatile("" ::: "memory");
      issueB(gg + 2, (gg + 2) % NBUF);
      compute(gg % NBUF);
    }
  }
  __builtin_amdgcn_s_waitcnt(p1_waitcnt(0));
  __syncthreads();

  // ---- epilogue, FAST PATH (round 6): a tile that lies wholly inside M x N, linear output rows -- every tile of the layers
  // that matter (M = 261 120 = 1020 x 256, N a multiple of the column tile).  The general path below spends ~3000 instructions per tile on
  // per-element predication (a scalar branch around each of the 128 stores, 256 row-validity selects in the BatchNorm partials), two ldexps per
  // element and the bias added three times -- as many as the K loop of a 512-deep layer (3900) and more than that of a 256-deep one (1950), with
  // the matrix pipe idle (SQ counters: MFMA busy 27 - 39 % of the CU-busy cycles).  Here: one ldexp by -(e_x + e_w) (exact; the
  // two-step ldexp stays for exponent sums outside +-120), the bias added once, partial sums without selects, 128 unconditional stores.
  // The arithmetic of every output and of the BatchNorm partials is the general path's (same operations in the same order).
  // FULL is a property of the LAUNCH (p1_launch: M a multiple of 256 when BatchNorm partials are asked for; any column count; linear or
  // gathered output rows), hence a
  // template parameter: with both epilogues in one kernel the register allocator kept copies of the 128 accumulator registers for either
  // arm and spilled 300 - 500 registers per thread in the 256-column form.
  if constexpr (FULL) {
    // (wave-uniform by construction; readfirstlane SAYS so: a branch the compiler takes for divergent keeps both arms' 128 accumulator
    //  registers alive)
    const int ew_u = __builtin_amdgcn_readfirstlane(p.wrec[1]), ex_u = __builtin_amdgcn_readfirstlane(ex);
    const int es = -(ex_u + ew_u);
    if (es >= -120 && es <= 120) {
      // (ONE ldexp, not a multiplication by 2^es: the compiler folds a multiplication into the fused multiply-adds of the BatchNorm partials
      //  below, keeps scaled AND unscaled accumulators alive and spills 256 registers per thread)
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = __builtin_ldexpf(acc[u][r], es);
    } else {
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = __builtin_ldexpf(__builtin_ldexpf(acc[u][r], -ex_u), -ew_u);
    }
    if (p.bias != nullptr) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int col = n0 + 32 * u + l31;
        const float b = col < p.N ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] += b;
      }
    }
    if (p.bn_part != nullptr)
      p1_full_tile_bn_partials<NT>((float*)smem, BN, l31, hh == 0, wave, acc, p.bn_part + (long long)tile_m * 3 * p.N, p.N, n0);
    constexpr unsigned OOBS = 0xFFFFFFF0u;      // a store / load at this offset is past the buffer's range: dropped / zero (columns past N)
    const bool accum = p.accumulate != 0;
    if (GEO && !p.geo.linear_out) {
      // gathered output rows (the parity classes of a strided backward-data pass): the byte offset of each accumulator row in a register,
      // the column added per store; one buffer resource over the whole output (below 4 GB: p1_launch)
      const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, (short)0, (int)(unsigned)p.y_bytes, 0x00020000);
      unsigned cb[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) cb[u] = n0 + 32 * u + l31 < p.N ? (unsigned)(n0 + 32 * u + l31) * 4u : OOBS;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rw = m0 + wave * 32 + 4 * hh + (r & 3) + 8 * (r >> 2);
        const int gb = rw / p.geo.HWg, rem = rw - gb * p.geo.HWg;
        const int gi = rem / p.geo.Wg, gj = rem - gi * p.geo.Wg;
        const unsigned ro = rw < p.M ? (unsigned)((gb * p.geo.Ho + gi * p.geo.so + p.geo.oy) * p.geo.Wo + gj * p.geo.so + p.geo.ox) * (unsigned)p.ldy * 4u : OOBS;
        float v[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) v[u] = acc[u][r];
        if (accum) {
          float o[NT];
#pragma unroll
          for (int u = 0; u < NT; ++u) o[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsY, (cb[u] == OOBS || ro == OOBS) ? OOBS : ro + cb[u], 0, 0));
#pragma unroll
          for (int u = 0; u < NT; ++u) v[u] += o[u];
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[u]), rsY, (cb[u] == OOBS || ro == OOBS) ? OOBS : ro + cb[u], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    // linear output rows: stores through a buffer resource over this tile's 256 rows -- one per-lane offset register per column tile (out of
    // range for columns past N), the row of accumulator element r in the scalar offset
    // (rows past M -- a ragged last row tile of a launch without BatchNorm partials -- lie past the resource's range: dropped)
    const unsigned rows_here = (unsigned)(p.M - m0 < 256 ? p.M - m0 : 256);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (long long)m0 * p.ldy + n0), (short)0,
                                                                        (int)(rows_here * (unsigned)p.ldy * 4u - (unsigned)n0 * 4u), 0x00020000);
    const int ld4 = p.ldy * 4;
    const int voff = (wave * 32 + 4 * hh) * ld4 + l31 * 4;
    int vo[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) vo[u] = n0 + 32 * u + l31 < p.N ? voff + 128 * u : (int)OOBS;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int soff = ((r & 3) + 8 * (r >> 2)) * ld4;
      float v[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) v[u] = acc[u][r];        // (copies: __builtin_bit_cast applied to a vector ELEMENT itself reads element 0 for every r -- hipcc 7.2)
      if (accum) {                                            // (the previous contents of the row, read before its stores: (acc + bias) + old, as the general path)
        float o[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) o[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsY, vo[u], soff, 0));
#pragma unroll
        for (int u = 0; u < NT; ++u) v[u] += o[u];
      }
#pragma unroll
      for (int u = 0; u < NT; ++u) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[u]), rsY, vo[u], soff, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }

  // ---- general path: ragged tiles, accumulation, gathered output rows
  // back to the operands' scale: 2^-(e_x + e_w) in two exact steps
  {
    const int ea = -ex, ew = -p.wrec[1];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = __builtin_ldexpf(__builtin_ldexpf(acc[u][r], ea), ew);
  }
  float bv[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int col = n0 + 32 * u + l31;
    bv[u] = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
  }
  const int rbase = m0 + wave * 32 + 4 * hh;
  if (p.bn_part != nullptr) {
    int colv[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) colv[u] = 32 * u + l31;
    cs_tile_bn_partials<NT, 16, 8, false>(
        (float*)smem, BN, colv, hh == 0, wave, min(256, p.M - m0),
        [&](int j, int i) { return acc[j][i] + bv[j]; },
        [&](int i) { return rbase + (i & 3) + 8 * (i >> 2) < p.M; }, p.bn_part + (long long)tile_m * 3 * p.N, p.N, n0);
  }
  long long orow[16];               // element offset of the output row of accumulator row r (GEO: the row's output pixel)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rw = rbase + (r & 3) + 8 * (r >> 2);
    if (GEO && !p.geo.linear_out) {
      const int rr = rw < p.M ? rw : 0;
      const int gb = rr / p.geo.HWg, rem = rr - gb * p.geo.HWg;
      const int gi = rem / p.geo.Wg, gj = rem - gi * p.geo.Wg;
      orow[r] = ((long long)(gb * p.geo.Ho + gi * p.geo.so + p.geo.oy) * p.geo.Wo + gj * p.geo.so + p.geo.ox) * p.ldy;
    } else {
      orow[r] = (long long)rw * p.ldy;
    }
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int col = n0 + 32 * u + l31;
    float add[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rw = rbase + (r & 3) + 8 * (r >> 2);
      add[r] = (p.accumulate && rw < p.M && col < p.N) ? p.y[orow[r] + col] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rw = rbase + (r & 3) + 8 * (r >> 2);
      if (rw < p.M && col < p.N) p.y[orow[r] + col] = (acc[u][r] + bv[u]) + add[r];
    }
  }
}

template <int NT, bool GEO, bool FULL>
__global__ __launch_bounds__(512) void p1_kernel(const P1Args p) {
  p1_body<NT, GEO, FULL>(p);
}
// the parity classes of a strided backward-data pass (catseg_gconv_bwd_data) as ONE launch: blockIdx.y = class.  A class of a 3 x 3 / 2 layer at
// 8 x 34 x 60 is 64 blocks on 256 CUs: four launches of a quarter of the chip each, one behind the other (round 6)
struct P1Multi { P1Args a[4]; };
template <int NT, bool FULL>
__global__ __launch_bounds__(512) void p1_multi_kernel(const P1Multi q) {
  p1_body<NT, true, FULL>(q.a[blockIdx.y]);
}

template <int NT>
void p1_launch_multi(const P1Multi& q, int n, hipStream_t st) {
  int blocks = 0;
  bool full = true;
  for (int i = 0; i < n; ++i) {
    const P1Args& a = q.a[i];
    const int b = a.ntn > 1 ? (a.tilesM + 7) / 8 * 8 * a.ntn : a.tilesM;
    blocks = b > blocks ? b : blocks;
    const bool gathered = !a.geo.linear_out;
    full = full && (gathered ? (a.y_bytes > 0 && a.y_bytes < 0xFFFFFFF0ull) : (a.M % 256 == 0 || a.bn_part == nullptr));
  }
  if (full) hipLaunchKernelGGL((p1_multi_kernel<NT, true>), dim3(blocks, n), dim3(512), 0, st, q);
  else hipLaunchKernelGGL((p1_multi_kernel<NT, false>), dim3(blocks, n), dim3(512), 0, st, q);
}
void p1_dispatch_multi(const P1Multi& q, int n, hipStream_t st) {
  switch (p1_bn(q.a[0].N) / 32) {
    case 1: p1_launch_multi<1>(q, n, st); break;
    case 2: p1_launch_multi<2>(q, n, st); break;
    case 3: p1_launch_multi<3>(q, n, st); break;
    case 4: p1_launch_multi<4>(q, n, st); break;
    case 5: p1_launch_multi<5>(q, n, st); break;
    case 6: p1_launch_multi<6>(q, n, st); break;
    case 7: p1_launch_multi<7>(q, n, st); break;
    default: p1_launch_multi<8>(q, n, st); break;
  }
}

template <int NT>
void p1_launch(const P1Args& a, bool geo, hipStream_t st) {
  const int per = 8 * a.ntn;
  const int blocks = a.ntn > 1 ? (a.tilesM + 7) / 8 * per : a.tilesM;
  // (ragged column counts ride on out-of-range store offsets; gathered output rows need 32-bit offsets over the whole output)
  const bool gathered = geo && !a.geo.linear_out;
  const bool full = gathered ? (a.y_bytes > 0 && a.y_bytes < 0xFFFFFFF0ull) : (a.M % 256 == 0 || a.bn_part == nullptr);
  if (geo) {
    if (full) hipLaunchKernelGGL((p1_kernel<NT, true, true>), dim3(blocks), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((p1_kernel<NT, true, false>), dim3(blocks), dim3(512), 0, st, a);
  } else {
    if (full) hipLaunchKernelGGL((p1_kernel<NT, false, true>), dim3(blocks), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((p1_kernel<NT, false, false>), dim3(blocks), dim3(512), 0, st, a);
  }
}

void p1_dispatch(const P1Args& a, bool geo, hipStream_t st) {
  switch (p1_bn(a.N) / 32) {
    case 1: p1_launch<1>(a, geo, st); break;
    case 2: p1_launch<2>(a, geo, st); break;
    case 3: p1_launch<3>(a, geo, st); break;
    case 4: p1_launch<4>(a, geo, st); break;
    case 5: p1_launch<5>(a, geo, st); break;
    case 6: p1_launch<6>(a, geo, st); break;
    case 7: p1_launch<7>(a, geo, st); break;
    default: p1_launch<8>(a, geo, st); break;
  }
}

// ---- backward-weight: dW[o][c] = sum_p dy[p][o] x[p][c] ------------------------------------------------------------------------------
// M = Cout (dy columns), N = Cin (x columns), reduction over the pixels: both operands are k-strided in memory.  A block owns a run of
// 32-pixel steps and an (MP x NP) tile of the result; per step the eight waves read the fp32 rows of dy and x coalesced (16 bytes per
// lane), split them in registers and write the two fp16 planes pixel-major into LDS; the MFMA fragments (8 consecutive pixels of one
// channel) come out of the hardware transpose ds_read_b64_tr_b16 (a 16-lane group turns 4 pixels x 16 channels around).  Each wave
// accumulates a (32 TM) x (32 TN) piece over the block's whole run; partial results go to slabs that p1t_reduce_kernel adds in a fixed
// order (deterministic).  Pixel row stride = 64 (mod 128) bytes: the four pixel rows one LDS cycle serves (64 bytes each: two 16-lane
// groups side by side) fall into four different 64-byte bank groups.
struct P1TArgs {
  const float* dy; int lddy; const unsigned* dyrec;
  const float* x; int ldx; const unsigned* xrec;
  float* slabs;                       // [gridDim.x][Mtot][Ntot] partial results (Mtot = nbm * MP, Ntot = nbn * NP)
  long long P;                        // pixels
  int M, N;                           // Cout, Cin
  int nbn, Mtot, Ntot;                // blockIdx.y = bm * nbn + bn
  int steps_per_block;                // 32-pixel steps per block
  unsigned long long dy_bytes, x_bytes;
  // gather form (kh x kw taps, stride, pad, dil): N = taps * Cin virtual columns (tap, c); pixel p = (image, yo, xo) of the Ho x Wo output
  int geo, Cin, kw, stride, pad, dil, Hi, Wi, Ho, Wo;
};

typedef short p1_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) p1_s16x4 p1_lds_s16x4;

template <int WM, int WN, int TM, int TN, bool GEO>
__global__ __launch_bounds__(512) void p1t_kernel(const P1TArgs p) {
  static_assert(WM * WN == 8, "eight waves");
  constexpr int MP = 32 * TM * WM, NP = 32 * TN * WN;
  constexpr int CH = MP + NP;                               // 16-bit columns of one LDS pixel row: dy channels, then x channels
  constexpr int RS = (CH * 2 + 63) / 128 * 128 + 64;        // row stride in bytes, = 64 (mod 128)
  constexpr int PL = 32 * RS;                               // one plane of a 32-pixel step
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * PL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int edy = p1_exponent(cs_amax_read(p.dyrec)), ex = p1_exponent(cs_amax_read(p.xrec));
  const int bm = blockIdx.y / p.nbn, bn = blockIdx.y - bm * p.nbn;
  const int mo = bm * MP, no = bn * NP;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, (short)0, (int)(unsigned)p.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)(unsigned)p.x_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  // staging: a step's 32 pixels x (MP + NP) channels in 4-float items.  Round 6: the dy quads and the x quads of a step are TWO item lists --
  // items 0 .. ITD - 1 walk the 32 x QM dy quads, items ITD .. ITEMS - 1 the 32 x QX x quads, consecutive lanes taking consecutive quads of
  // one pixel row (coalesced) -- so that an item's operand is known at compile time: ONE load per item (one list over the MP + NP columns
  // needed a dy load AND an x load per item, one of them out of range, and a select), no per-lane operand selects in the split either.
  constexpr int QM = MP / 4, QT = (MP + NP) / 4, QX = QT - QM;
  constexpr int ITD = (32 * QM + 511) / 512, ITX = (32 * QX + 511) / 512, ITEMS = ITD + ITX;
  const long long step0 = (long long)blockIdx.x * p.steps_per_block;
  f32x4 raw[ITEMS];
  // (pixel, column quad of the LDS row, live) of item i; returns whether it is an x item
  auto item = [&](const int i, int& px, int& c4, bool& live) -> bool {
    const bool isx = i >= ITD;
    const int q = (isx ? i - ITD : i) * 512 + tid, Q = isx ? QX : QM;
    px = q / Q;
    c4 = (isx ? QM : 0) + (q - px * Q);
    live = q < 32 * Q;
    return isx;
  };
  // gather form: what does not change from step to step is decoded ONCE per item -- the tap and channel of its column quad, packed
  // ky | kx << 8 | c << 16 (-1: a dy item, or a column past N)
  int itap[ITEMS];
  const float r_hw = GEO ? 1.f / (float)(p.Ho * p.Wo) : 0.f, r_wo = GEO ? 1.f / (float)p.Wo : 0.f;
  if (GEO) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      int px, c4;
      bool live;
      const bool isx = item(i, px, c4, live);
      itap[i] = -1;
      if (live && isx && no + (c4 - QM) * 4 < p.N) {
        const int v = no + (c4 - QM) * 4, tap = v / p.Cin, c = v - tap * p.Cin;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        itap[i] = ky | (kx << 8) | (c << 16);
      }
    }
  }
  auto fdiv = [](int n, int d, float rd) -> int {      // n / d for 0 <= n < 2^23 (exact in float), one correction step each way
    int q = (int)((float)n * rd);
    q -= (int)(q * d > n);
    q += (int)((q + 1) * d <= n);
    return q;
  };
  // plain (not gathered) operands: an item's byte offset advances by a CONSTANT per 32-pixel step -- computed once (round 6: the 64-bit
  // address arithmetic of every item of every step was most of the kernel's 7 VALU instructions per MFMA; profiles/r06_pmc_sq_p1t_fpixel.json:
  // matrix pipes busy 18 % of the CU-busy cycles at 2.3 GHz, 62 % of the wave cycles waiting).
  unsigned off0[ITEMS], stepb[ITEMS];
  int pxv[ITEMS];
  if (!GEO) {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      int px, c4;
      bool live;
      const bool isx = item(i, px, c4, live);
      const bool ok = live && (isx ? (no + (c4 - QM) * 4 < p.N) : (mo + c4 * 4 < p.M));
      const long long pix0 = step0 * 32 + px;
      off0[i] = !ok ? OOB : (isx ? (unsigned)((pix0 * p.ldx + no + (c4 - QM) * 4) * 4) : (unsigned)((pix0 * p.lddy + mo + c4 * 4) * 4));
      stepb[i] = (unsigned)(32 * (isx ? p.ldx : p.lddy) * 4);
      pxv[i] = ok ? px : (1 << 30);
    }
  }
  auto load_plain = [&](int srel) {          // step step0 + srel
    const long long left = p.P - (step0 + srel) * 32;          // pixels of this step that exist
    const int lim = left > 32 ? 32 : (int)left;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const unsigned off = pxv[i] < lim ? off0[i] + (unsigned)srel * stepb[i] : OOB;
      raw[i] = i >= ITD ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0))
                        : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, off, 0, 0));
    }
  };
  // gather form, INCREMENTAL (round 6): the pixel (image, yo, xo) of an x item advances by 32 pixels per step -- two conditional wraps of xo
  // (valid for Wo >= 16: xo + 32 < 3 Wo) and one of yo instead of two divisions per item and step; the tap's offsets (ky dil - pad,
  // kx dil - pad) and channel are per-item constants; 32-bit offsets (the operands are below 4 GB).  dy items advance by a constant.  load()
  // is called for consecutive steps in order, which is what the running state relies on.  Small maps keep the per-step decode.
  const bool geo_inc = GEO && p.Wo >= 16 && p.Ho >= 2;
  int gb[GEO ? ITEMS : 1], gy[GEO ? ITEMS : 1], gx[GEO ? ITEMS : 1], gcy[GEO ? ITEMS : 1], gcx[GEO ? ITEMS : 1], gcc[GEO ? ITEMS : 1];
  if (geo_inc) {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      int px, c4;
      bool live;
      const bool isx = item(i, px, c4, live);
      const long long pix0 = step0 * 32 + px;
      // dy items: plain offsets (off0 / stepb / pxv as in the plain form); x items: running pixel
      const bool okd = live && !isx && mo + c4 * 4 < p.M;
      off0[i] = okd ? (unsigned)((pix0 * p.lddy + mo + c4 * 4) * 4) : OOB;
      stepb[i] = (unsigned)(32 * p.lddy * 4);
      pxv[i] = okd ? px : (1 << 30);
      const int b = (int)(pix0 / hw), rem = (int)(pix0 - (long long)b * hw);
      gb[i] = b; gy[i] = rem / p.Wo; gx[i] = rem - gy[i] * p.Wo;
      gcy[i] = (itap[i] & 255) * p.dil - p.pad; gcx[i] = ((itap[i] >> 8) & 255) * p.dil - p.pad; gcc[i] = itap[i] >> 16;
    }
  }
  const int n_img = GEO ? (int)(p.P / ((long long)p.Ho * p.Wo)) : 0;
  auto load_geo_inc = [&](int srel) {
    const long long left = p.P - (step0 + srel) * 32;
    const int lim = left > 32 ? 32 : (int)left;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (i < ITD) {
        const unsigned offd = pxv[i] < lim ? off0[i] + (unsigned)srel * stepb[i] : OOB;
        raw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, offd, 0, 0));
      } else if (GEO) {
        const int yi = gy[i] * p.stride + gcy[i], xi = gx[i] * p.stride + gcx[i];
        const bool ok = itap[i] >= 0 && gb[i] < n_img && (unsigned)yi < (unsigned)p.Hi && (unsigned)xi < (unsigned)p.Wi;
        const unsigned offx = ok ? ((unsigned)((gb[i] * p.Hi + yi) * p.Wi + xi) * (unsigned)p.ldx + (unsigned)gcc[i]) * 4u : OOB;
        // next step: 32 pixels on
        int x2 = gx[i] + 32, y2 = gy[i];
        bool c = x2 >= p.Wo; x2 -= c ? p.Wo : 0; y2 += c;
        c = x2 >= p.Wo; x2 -= c ? p.Wo : 0; y2 += c;
        c = y2 >= p.Ho; y2 -= c ? p.Ho : 0;
        gx[i] = x2; gy[i] = y2; gb[i] += c;
        raw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, offx, 0, 0));
      }
    }
  };
  auto load = [&](long long s) {
    if (!GEO) {
      load_plain((int)(s - step0));
      return;
    }
    if (geo_inc) {
      load_geo_inc((int)(s - step0));
      return;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      int px, c4;
      bool live;
      const bool isx = item(i, px, c4, live);
      const long long pix = s * 32 + px;
      unsigned off = OOB;
      if (live && pix < p.P) {
        if (!isx) {
          if (mo + c4 * 4 < p.M) off = (unsigned)(pix * p.lddy + mo + c4 * 4) * 4u;
        } else if (itap[i] >= 0) {
          const int hw = p.Ho * p.Wo, b = fdiv((int)pix, hw, r_hw), rem = (int)pix - b * hw;
          const int yo = fdiv(rem, p.Wo, r_wo), xo = rem - yo * p.Wo;
          const int yi = yo * p.stride - p.pad + (itap[i] & 255) * p.dil, xi = xo * p.stride - p.pad + ((itap[i] >> 8) & 255) * p.dil;
          if ((unsigned)yi < (unsigned)p.Hi && (unsigned)xi < (unsigned)p.Wi)
            off = (unsigned)(((long long)(b * p.Hi + yi) * p.Wi + xi) * p.ldx + (itap[i] >> 16)) * 4u;
        }
      }
      raw[i] = isx ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0))
                   : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, off, 0, 0));
    }
  };
  auto stash = [&](int buf) {
    char* base = smem + buf * 2 * PL;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      int px, c4;
      bool live;
      const bool isx = item(i, px, c4, live);
      if (live) {
        const int e = isx ? ex : edy;
        const f32x4 v = raw[i];
        u16 h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xs = __builtin_ldexpf(v[j], e);
          const _Float16 hv = (_Float16)xs;
          h[j] = __builtin_bit_cast(u16, hv);
          l[j] = __builtin_bit_cast(u16, (_Float16)(xs - (float)hv));
        }
        char* d = base + px * RS + c4 * 8;
        *(unsigned long long*)d = (unsigned long long)h[0] | ((unsigned long long)h[1] << 16) | ((unsigned long long)h[2] << 32) | ((unsigned long long)h[3] << 48);
        *(unsigned long long*)(d + PL) = (unsigned long long)l[0] | ((unsigned long long)l[1] << 16) | ((unsigned long long)l[2] << 32) | ((unsigned long long)l[3] << 48);
      }
    }
  };
  // fragment of a 32-column tile: lane (j = lane & 31, g = lane >> 5) needs pixels 16 kt + 8 g .. + 8 of column col0 + j.  The 16-lane
  // group of the lane (columns col0 + 16 (j >> 4) .. + 16) transposes 4 pixels x 16 columns per read: lane li of the group supplies the
  // address of (pixel li >> 2 of the four, columns 4 (li & 3) .. + 4) and receives the four pixels of column li
  const int fr_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * RS + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  auto frag = [&](const char* plane, int col0, int kt) -> half8 {
    const char* src = plane + fr_off + 16 * kt * RS + col0 * 2;
    const p1_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((p1_lds_s16x4*)src);
    const p1_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((p1_lds_s16x4*)(src + 4 * RS));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(half8, v8);
  };

  const int nsteps = p.steps_per_block;
  load(step0);
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    stash(buf);                                     // (waits for the loads of step s; buffer s & 1 was last read in step s - 2: behind the barrier of step s - 1)
    if (s + 1 < nsteps) load(step0 + s + 1);
    __syncthreads();
    const char* hp = smem + buf * 2 * PL;
    const char* lp = hp + PL;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = frag(hp, 32 * (wm * TM + i), kt);
        al[i] = frag(lp, 32 * (wm * TM + i), kt);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = frag(hp, MP + 32 * (wn * TN + j), kt);
        bl[j] = frag(lp, MP + 32 * (wn * TN + j), kt);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
    }
  }
  // this block's tile of its slab, scaled back
  float* slab = p.slabs + (long long)blockIdx.x * p.Mtot * p.Ntot;
  const int l31 = lane & 31, hh = lane >> 5;
  const int sa = -edy, sb = -ex;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rw = mo + 32 * (wm * TM + i) + (r & 3) + 8 * (r >> 2) + 4 * hh, col = no + 32 * (wn * TN + j) + l31;
        slab[(long long)rw * p.Ntot + col] = __builtin_ldexpf(__builtin_ldexpf(acc[i][j][r], sa), sb);
      }
}

// dw[o][c] = sum over slabs, fixed order.  Block = 64 outputs x 4 slab groups (group g takes slabs g, g + 4, ...), eight independent load chains
// per thread, the groups merged through LDS: the 256 slabs of a 64 x 256 layer were 64 dependent rounds of four loads (19 us, latency-bound;
// 54 launches per HRNet-W48 step), now 8 rounds
__global__ __launch_bounds__(256) void p1t_reduce_kernel(const float* __restrict__ slabs, int nslabs, int Mtot, int Ntot, int M, int N, float* __restrict__ dw) {
  __shared__ float red[4][64];
  const int t = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + t;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < M * N) {
    const int o = i / N, c = i - o * N;
    const float* s = slabs + (long long)o * Ntot + c;
    const long long stride = (long long)Mtot * Ntot;
    int k0 = grp;
    for (; k0 + 28 < nslabs; k0 += 32) {      // (eight UNCONDITIONAL loads: a predicate around each one serialises their round trips)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s[(long long)(k0 + 4 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += v[u];
    }
    for (; k0 < nslabs; k0 += 4) a[0] += s[(long long)k0 * stride];
  }
  red[grp][t] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (grp == 0 && i < M * N) dw[i] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

struct P1TPlan { int kind, MP, NP; };
// block tile of the result by shape: kind -> (WM, WN, TM, TN); at most 128 accumulator registers per wave
P1TPlan p1t_plan(int M, int N, bool geo = false) {
  if (M <= 0 || N <= 0 || M > 1024 || N > (geo ? 16384 : 1024)) return {0, 0, 0};
  if (M <= 64 && N <= 128) return {8, 64, 128};         // 2 x 4 waves of 32 x 32
  if (M <= 64) return {1, 64, 256};                     // 2 x 4 waves of 32 x 64
  if (N <= 64) return {2, 256, 64};                     // 4 x 2 waves of 64 x 32
  if (M <= 128 && N <= 128) return {3, 128, 128};       // 2 x 4 waves of 64 x 32
  if (M <= 128 || geo) return {9, 128, 256};            // 2 x 4 waves of 64 x 64 (gather form: the 256 x 256 tile would spill)
  return {4, 256, 256};                                 // 2 x 4 waves of 128 x 64
}

int g_p1t_blocks = 256;

}  // namespace

extern "C" int catseg_pconv1_supported(int N, int K) {
  return (N >= 32 && K >= 32 && K % 8 == 0 && N <= 2048 && K <= 8160) ? 1 : 0;       // (K < 8192: the 32-bit reciprocal of the tap decode)
}

extern "C" size_t catseg_pconv1_wimg_bytes(int N, int K) {
  return (size_t)p1_groups(K) * p1_ntn(N) * 4 * p1_bn(N) * 32;
}

// entries: DEVICE array of n records {int64 weight offset (floats, relative to flat), int64 image offset (bytes, relative to wimg_base),
// int32 O, int32 I, int32 transposed, int32 pad}: the image of (N = O, K = I), or with transposed != 0 of (N = I, K = O) (backward-data);
// records = n x {uint32 bits of max|w|, int32 exponent} (DEVICE): zeroed, filled by the amax launch, completed by the image launch
extern "C" int catseg_pconv1_prep_batch(const float* flat, int n, const void* entries, void* wimg_base, void* records, catseg_stream_t stream) {
  CS_REQUIRE(flat && entries && wimg_base && records && n > 0, "pconv1 prep batch: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(records, 0, (size_t)n * 8, st) != hipSuccess) { catseg_set_error("pconv1 prep: memset failed"); return CATSEG_EHIP; }
  hipLaunchKernelGGL(p1_amax_batch_kernel, dim3(8, n), dim3(256), 0, st, flat, (const P1Entry*)entries, (unsigned*)records);
  hipLaunchKernelGGL(p1_prep_batch_kernel, dim3(32, n), dim3(256), 0, st, flat, (const P1Entry*)entries, (unsigned char*)wimg_base, (unsigned*)records);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// y[M][N] (+)= x[M][K] . B^T (+ bias) with B = the weight image of (N, K); x_rec: the amax record of x (its producer's); w_rec: the
// image's {amax bits, exponent}.  bn_part may be null; otherwise per-(256-row tile, channel) BatchNorm partials as catseg_conv2d_fwd_bnstats.
extern "C" int catseg_pconv1(long long M, int N, int K, const float* x, int ldx, const void* x_rec, const void* wimg, const void* w_rec,
                             const float* bias, float* y, int ldy, int accumulate, float* bn_part, size_t bn_part_floats, int* tile_rows,
                             int* n_tiles, catseg_stream_t stream) {
  CS_REQUIRE(M > 0 && catseg_pconv1_supported(N, K) && x && x_rec && wimg && w_rec && y, "pconv1: unsupported shape or null argument");
  CS_REQUIRE(ldx >= K && ldx % 4 == 0 && ldy >= N && cs_aligned16(x) && cs_aligned16(wimg) && ((uintptr_t)y & 3) == 0, "pconv1: alignment / row strides");
  CS_REQUIRE((unsigned long long)M * (unsigned long long)ldx * 4ull < 0xFFFFFFF0ull && M < (1ll << 31) - 256, "pconv1: activation beyond one 4 GB buffer resource");
  P1Args a = {};
  a.x = x; a.ldx = ldx; a.xrec = (const unsigned*)x_rec; a.wimg = (const unsigned char*)wimg; a.wrec = (const int*)w_rec;
  a.y = y; a.ldy = ldy; a.bias = bias; a.M = (int)M; a.N = N; a.K = K;
  a.tilesM = (int)((M + 255) / 256); a.ntn = p1_ntn(N); a.accumulate = accumulate;
  a.x_bytes = (unsigned long long)M * ldx * 4ull; a.img_bytes = catseg_pconv1_wimg_bytes(N, K);
  if (bn_part != nullptr) {
    CS_REQUIRE(tile_rows && n_tiles, "pconv1: tile_rows / n_tiles");
    *tile_rows = 0; *n_tiles = 0;
    if ((size_t)a.tilesM * 3 * N <= bn_part_floats) {
      a.bn_part = bn_part;
      *tile_rows = 256; *n_tiles = a.tilesM;
    }
  }
  p1_dispatch(a, false, (hipStream_t)stream);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_pconv1_wgrad_supported(int Cout, int Cin) { return p1t_plan(Cout, Cin).kind != 0 && Cout % 4 == 0 && Cin % 4 == 0 ? 1 : 0; }

namespace {
struct P1TGrid { P1TPlan pl; int nbm, nbn, Mtot, Ntot, blocks, steps_per_block; };
P1TGrid p1t_grid(long long P, int Cout, int Cin, bool geo = false) {
  P1TGrid g = {};
  g.pl = p1t_plan(Cout, Cin, geo);
  if (!g.pl.kind) return g;
  g.nbm = (Cout + g.pl.MP - 1) / g.pl.MP; g.nbn = (Cin + g.pl.NP - 1) / g.pl.NP;
  g.Mtot = g.nbm * g.pl.MP; g.Ntot = g.nbn * g.pl.NP;
  const long long steps = (P + 31) / 32;
  long long want = g_p1t_blocks / (g.nbm * g.nbn);
  if (want < 1) want = 1;
  if (want > steps) want = steps;
  g.steps_per_block = (int)((steps + want - 1) / want);
  g.blocks = (int)((steps + g.steps_per_block - 1) / g.steps_per_block);
  return g;
}
}  // namespace

namespace {
template <bool GEO>
void p1t_launch(const P1TArgs& a, const P1TGrid& g, hipStream_t st) {
  const dim3 grid(g.blocks, g.nbm * g.nbn);
  switch (g.pl.kind) {
    case 8: hipLaunchKernelGGL((p1t_kernel<2, 4, 1, 1, GEO>), grid, dim3(512), 0, st, a); break;
    case 1: hipLaunchKernelGGL((p1t_kernel<2, 4, 1, 2, GEO>), grid, dim3(512), 0, st, a); break;
    case 2: hipLaunchKernelGGL((p1t_kernel<4, 2, 2, 1, GEO>), grid, dim3(512), 0, st, a); break;
    case 3: hipLaunchKernelGGL((p1t_kernel<2, 4, 2, 1, GEO>), grid, dim3(512), 0, st, a); break;
    case 9: hipLaunchKernelGGL((p1t_kernel<2, 4, 2, 2, GEO>), grid, dim3(512), 0, st, a); break;
    default: hipLaunchKernelGGL((p1t_kernel<2, 4, 4, 2, GEO>), grid, dim3(512), 0, st, a); break;
  }
}
void p1t_dispatch(const P1TArgs& a, const P1TGrid& g, bool geo, hipStream_t st) {
  if (geo) p1t_launch<true>(a, g, st);
  else p1t_launch<false>(a, g, st);
}
}  // namespace

extern "C" size_t catseg_pconv1_wgrad_workspace(long long P, int Cout, int Cin) {
  const P1TGrid g = p1t_grid(P, Cout, Cin);
  if (!g.pl.kind) return 0;
  return (size_t)g.blocks * g.Mtot * g.Ntot * 4;
}

extern "C" int catseg_debug_set_pconv1_wgrad_blocks(int n) { g_p1t_blocks = n > 0 ? n : 256; return CATSEG_OK; }

// dw[Cout][Cin] = dy^T . x over P pixels; both operands fp32 rows with their producers' amax records
extern "C" int catseg_pconv1_wgrad(long long P, int Cout, int Cin, const float* dy, int lddy, const void* dy_rec, const float* x, int ldx,
                                   const void* x_rec, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  const P1TGrid g = p1t_grid(P, Cout, Cin);
  CS_REQUIRE(P > 0 && g.pl.kind && Cout % 4 == 0 && Cin % 4 == 0 && dy && x && dy_rec && x_rec && dw && workspace, "pconv1 wgrad: unsupported shape or null argument");
  CS_REQUIRE(lddy >= Cout && ldx >= Cin && lddy % 4 == 0 && ldx % 4 == 0 && cs_aligned16(dy) && cs_aligned16(x) && cs_aligned16(workspace),
             "pconv1 wgrad: alignment / row strides");
  CS_REQUIRE((unsigned long long)P * lddy * 4ull < 0xFFFFFFF0ull && (unsigned long long)P * ldx * 4ull < 0xFFFFFFF0ull, "pconv1 wgrad: operand beyond 4 GB");
  CS_REQUIRE(workspace_bytes >= catseg_pconv1_wgrad_workspace(P, Cout, Cin), "pconv1 wgrad: workspace too small");
  P1TArgs a = {};
  a.dy = dy; a.lddy = lddy; a.dyrec = (const unsigned*)dy_rec; a.x = x; a.ldx = ldx; a.xrec = (const unsigned*)x_rec;
  a.slabs = (float*)workspace; a.P = P; a.M = Cout; a.N = Cin; a.nbn = g.nbn; a.Mtot = g.Mtot; a.Ntot = g.Ntot;
  a.steps_per_block = g.steps_per_block;
  a.dy_bytes = (unsigned long long)P * lddy * 4ull; a.x_bytes = (unsigned long long)P * ldx * 4ull;
  hipStream_t st = (hipStream_t)stream;
  p1t_dispatch(a, g, false, st);
  hipLaunchKernelGGL(p1t_reduce_kernel, dim3((Cout * Cin + 63) / 64), dim3(256), 0, st, (const float*)workspace, g.blocks, g.Mtot, g.Ntot, Cout, Cin, dw);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}


// =====================================================================================================================================
// "gather" launches of the same two kernels: dense convolutions with kh x kw taps, stride, padding and dilation whose input carries an amax
// record and that neither the direct 3x3 kernels (Cin = Cout, stride 1) nor the blocked-plane kernels (K >= 2048 with >= 192 columns) take:
// the 3x3 / stride 2 layers of the HRNet fuse chains and transitions (models/HRNetv2.py:237-261, :176-198 of the reference), the
// 256 -> 48 transition, the second stem convolution.  A lane's 8-deep k run lies inside ONE tap (Cin % 8 == 0), so its 32 bytes are still
// contiguous in the NHWC row of the tap's input pixel; padding taps are out-of-range offsets (zeros).  Backward-data of a stride-s layer =
// one launch per input-pixel parity class with that class's sub-grid of taps (no product with a structural zero), writing every s-th pixel.
namespace {
unsigned p1_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

bool g1_shape_ok(const catseg_conv_desc* d) {
  return d && !d->stem4 && d->groups <= 1 && d->Cin % 8 == 0 && d->Cout % 4 == 0 && d->kh >= 1 && d->kw >= 1 && d->kh * d->kw <= 25 &&
         d->stride >= 1 && d->stride <= 2 && d->kh >= d->stride && d->kw >= d->stride && (d->stride == 1 || d->dil == 1);
}

// sub-grid of the taps that reach input-pixel parity py (one dimension): k = k0 + a * ks, a < nk; y0 = (py + pad - k0 dil) / s
void g1_class(int k, int s, int pad, int dil, int py, int* k0, int* ks, int* nk, int* y0) {
  *ks = s; *k0 = -1; *nk = 0; *y0 = 0;            // (dil == 1 whenever s > 1)
  for (int kk = 0; kk < k; ++kk)
    if ((py + pad - kk * dil) % s == 0 && py + pad - kk * dil >= -1000000) {
      if (*k0 < 0) { *k0 = kk; *y0 = (py + pad - kk * dil) / s; }
      ++*nk;
      if (s == 1) break;
    }
  if (s == 1) { *k0 = 0; *ks = 1; *nk = k; *y0 = pad; }
}
}  // namespace

extern "C" int catseg_gconv_supported(const catseg_conv_desc* d) {
  if (!g1_shape_ok(d)) return 0;
  return (catseg_pconv1_supported(d->Cout, d->kh * d->kw * d->Cin) && d->kh * d->kw * d->Cin <= 8192 - 32 && d->Cin <= 1024 && d->Cout <= 512) ? 1 : 0;
}

// bytes of the forward image (backward_data = 0) or of ALL class images of backward-data (class c = py * stride + px at offset
// c * catseg_gconv_class_bytes)
extern "C" size_t catseg_gconv_class_bytes(const catseg_conv_desc* d) {
  // every class image is sized for the largest class (all taps of one parity in both dimensions)
  const int s = d->stride, nky = (d->kh + s - 1) / s, nkx = (d->kw + s - 1) / s;
  return catseg_pconv1_wimg_bytes(d->Cin, nky * nkx * d->Cout);
}
extern "C" size_t catseg_gconv_wimg_bytes(const catseg_conv_desc* d, int backward_data) {
  if (!backward_data) return catseg_pconv1_wimg_bytes(d->Cout, d->kh * d->kw * d->Cin);
  return (size_t)d->stride * d->stride * catseg_gconv_class_bytes(d);
}

// fills the host-side prep entries of a layer: entries[0] (forward) or entries[0 .. s*s) (backward-data classes); returns the count.
// entry layout = P1Entry (64 bytes); w_off / img_off are written relative to the caller's bases (img_off: base + class offset)
extern "C" int catseg_gconv_entries(const catseg_conv_desc* d, int backward_data, long long w_off, long long img_off, void* entries_out) {
  if (!g1_shape_ok(d)) return 0;
  P1Entry* e = (P1Entry*)entries_out;
  if (!backward_data) {
    e[0] = P1Entry{w_off, img_off, d->Cout, d->Cin, d->kh, d->kw, 0, 0, 1, d->kh, 0, 1, d->kw, 0};
    return 1;
  }
  const int s = d->stride;
  const size_t cb = catseg_gconv_class_bytes(d);
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px) {
      int ky0, kys, nky, y0, kx0, kxs, nkx, x0;
      g1_class(d->kh, s, d->pad, d->dil, py, &ky0, &kys, &nky, &y0);
      g1_class(d->kw, s, d->pad, d->dil, px, &kx0, &kxs, &nkx, &x0);
      e[py * s + px] = P1Entry{w_off, img_off + (long long)((py * s + px) * cb), d->Cout, d->Cin, d->kh, d->kw, 1, ky0, kys, nky, kx0, kxs, nkx, 0};
    }
  return s * s;
}

// y = conv(x, w) (+ bias) -- forward gather launch; wimg = forward image, BatchNorm partials as catseg_conv2d_fwd_bnstats
extern "C" int catseg_gconv_fwd(const catseg_conv_desc* d, const float* x, const void* x_rec, const void* wimg, const void* w_rec,
                                const float* bias, float* y, float* bn_part, size_t bn_part_floats, int* tile_rows, int* n_tiles,
                                catseg_stream_t stream) {
  CS_REQUIRE(catseg_gconv_supported(d) && x && x_rec && wimg && w_rec && y, "gconv fwd: unsupported shape or null argument");
  CS_REQUIRE(d->ldx >= d->Cin && d->ldx % 4 == 0 && d->ldy >= d->Cout && cs_aligned16(x) && cs_aligned16(wimg), "gconv fwd: alignment / row strides");
  const long long rows_in = (long long)d->B * d->H * d->W, M = (long long)d->B * d->Ho * d->Wo;
  CS_REQUIRE((unsigned long long)rows_in * d->ldx * 4ull < 0xFFFFFFF0ull && M < (1ll << 31) - 256, "gconv fwd: activation beyond one 4 GB buffer resource");
  P1Args a = {};
  a.x = x; a.ldx = d->ldx; a.xrec = (const unsigned*)x_rec; a.wimg = (const unsigned char*)wimg; a.wrec = (const int*)w_rec;
  a.y = y; a.ldy = d->ldy; a.bias = bias; a.M = (int)M; a.N = d->Cout; a.K = d->kh * d->kw * d->Cin;
  a.tilesM = (int)((M + 255) / 256); a.ntn = p1_ntn(a.N);
  a.x_bytes = (unsigned long long)rows_in * d->ldx * 4ull; a.img_bytes = catseg_pconv1_wimg_bytes(a.N, a.K);
  P1Geo& g = a.geo;
  g.Wg = d->Wo; g.HWg = d->Ho * d->Wo; g.si = d->stride; g.y0 = -d->pad; g.x0 = -d->pad; g.ya = d->dil; g.xa = d->dil; g.nb = d->kw;
  g.Hi = d->H; g.Wi = d->W; g.C = d->Cin; g.mC = p1_magic(d->Cin); g.mnb = p1_magic(d->kw);
  g.so = 1; g.oy = 0; g.ox = 0; g.Ho = d->Ho; g.Wo = d->Wo; g.linear_out = 1;
  if (bn_part != nullptr) {
    CS_REQUIRE(tile_rows && n_tiles, "gconv fwd: tile_rows / n_tiles");
    *tile_rows = 0; *n_tiles = 0;
    if ((size_t)a.tilesM * 3 * a.N <= bn_part_floats) {
      a.bn_part = bn_part;
      *tile_rows = 256; *n_tiles = a.tilesM;
    }
  }
  p1_dispatch(a, true, (hipStream_t)stream);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// dx (+)= conv_transpose(dy, w): wimg_classes = the stride^2 class images (catseg_gconv_entries(backward_data = 1)), one launch per class
extern "C" int catseg_gconv_bwd_data(const catseg_conv_desc* d, const float* dy, const void* dy_rec, const void* wimg_classes, const void* w_rec,
                                     float* dx, int accumulate, catseg_stream_t stream) {
  CS_REQUIRE(g1_shape_ok(d) && dy && dy_rec && wimg_classes && w_rec && dx, "gconv bwd_data: unsupported shape or null argument");
  CS_REQUIRE(d->ldy >= d->Cout && d->ldy % 4 == 0 && d->ldx >= d->Cin && cs_aligned16(dy) && cs_aligned16(wimg_classes), "gconv bwd_data: alignment / row strides");
  const long long rows_dy = (long long)d->B * d->Ho * d->Wo;
  CS_REQUIRE((unsigned long long)rows_dy * d->ldy * 4ull < 0xFFFFFFF0ull && (long long)d->B * d->H * d->W < (1ll << 31) - 256, "gconv bwd_data: operand beyond 4 GB");
  const int s = d->stride;
  const size_t cb = catseg_gconv_class_bytes(d);
  P1Multi multi;
  int nmulti = 0;
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px) {
      int ky0, kys, nky, y0, kx0, kxs, nkx, x0;
      g1_class(d->kh, s, d->pad, d->dil, py, &ky0, &kys, &nky, &y0);
      g1_class(d->kw, s, d->pad, d->dil, px, &kx0, &kxs, &nkx, &x0);
      const int Hg = (d->H - py + s - 1) / s, Wg = (d->W - px + s - 1) / s;
      if (Hg <= 0 || Wg <= 0) continue;
      CS_REQUIRE(nky > 0 && nkx > 0, "gconv bwd_data: a parity class without taps");
      P1Args a = {};
      a.x = dy; a.ldx = d->ldy; a.xrec = (const unsigned*)dy_rec;
      a.wimg = (const unsigned char*)wimg_classes + (size_t)(py * s + px) * cb; a.wrec = (const int*)w_rec;
      a.y = dx; a.ldy = d->ldx; a.bias = nullptr; a.M = d->B * Hg * Wg; a.N = d->Cin; a.K = nky * nkx * d->Cout;
      a.tilesM = (a.M + 255) / 256; a.ntn = p1_ntn(a.N); a.accumulate = accumulate;
      a.x_bytes = (unsigned long long)rows_dy * d->ldy * 4ull; a.img_bytes = catseg_pconv1_wimg_bytes(a.N, a.K);
      a.y_bytes = (unsigned long long)d->B * d->H * d->W * (unsigned long long)d->ldx * 4ull;
      P1Geo& g = a.geo;
      g.Wg = Wg; g.HWg = Hg * Wg; g.si = 1; g.y0 = y0; g.x0 = x0;
      g.ya = s == 1 ? -d->dil : -1; g.xa = g.ya; g.nb = nkx;
      g.Hi = d->Ho; g.Wi = d->Wo; g.C = d->Cout; g.mC = p1_magic(d->Cout); g.mnb = p1_magic(nkx);
      g.so = s; g.oy = py; g.ox = px; g.Ho = d->H; g.Wo = d->W; g.linear_out = (s == 1) ? 1 : 0;
      if (s == 2) multi.a[nmulti++] = a;      // (the four classes of a stride-2 layer: one launch below; every class has the same column count)
      else p1_dispatch(a, true, (hipStream_t)stream);
    }
  if (nmulti == 1) p1_dispatch(multi.a[0], true, (hipStream_t)stream);
  else if (nmulti > 1) p1_dispatch_multi(multi, nmulti, (hipStream_t)stream);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_gconv_wgrad_supported(const catseg_conv_desc* d) {
  return (g1_shape_ok(d) && p1t_plan(d->Cout, d->kh * d->kw * d->Cin, true).kind != 0 && d->Cin % 4 == 0) ? 1 : 0;
}
extern "C" size_t catseg_gconv_wgrad_workspace(const catseg_conv_desc* d) {
  const P1TGrid g = p1t_grid((long long)d->B * d->Ho * d->Wo, d->Cout, d->kh * d->kw * d->Cin, true);
  return g.pl.kind ? (size_t)g.blocks * g.Mtot * g.Ntot * 4 : 0;
}

// dw[Cout][kh][kw][Cin] = sum over output pixels of dy (x) x(gathered): the OHWI layout IS [Cout][taps * Cin]
extern "C" int catseg_gconv_bwd_weight(const catseg_conv_desc* d, const float* dy, const void* dy_rec, const float* x, const void* x_rec,
                                       float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(catseg_gconv_wgrad_supported(d) && dy && x && dy_rec && x_rec && dw && workspace, "gconv wgrad: unsupported shape or null argument");
  const long long P = (long long)d->B * d->Ho * d->Wo, rows_in = (long long)d->B * d->H * d->W;
  const int N = d->kh * d->kw * d->Cin;
  const P1TGrid g = p1t_grid(P, d->Cout, N, true);
  CS_REQUIRE(d->ldy >= d->Cout && d->ldx >= d->Cin && d->ldy % 4 == 0 && d->ldx % 4 == 0 && cs_aligned16(dy) && cs_aligned16(x) && cs_aligned16(workspace),
             "gconv wgrad: alignment / row strides");
  CS_REQUIRE((unsigned long long)P * d->ldy * 4ull < 0xFFFFFFF0ull && (unsigned long long)rows_in * d->ldx * 4ull < 0xFFFFFFF0ull, "gconv wgrad: operand beyond 4 GB");
  CS_REQUIRE(workspace_bytes >= catseg_gconv_wgrad_workspace(d), "gconv wgrad: workspace too small");
  P1TArgs a = {};
  a.dy = dy; a.lddy = d->ldy; a.dyrec = (const unsigned*)dy_rec; a.x = x; a.ldx = d->ldx; a.xrec = (const unsigned*)x_rec;
  a.slabs = (float*)workspace; a.P = P; a.M = d->Cout; a.N = N; a.nbn = g.nbn; a.Mtot = g.Mtot; a.Ntot = g.Ntot;
  a.steps_per_block = g.steps_per_block;
  a.dy_bytes = (unsigned long long)P * d->ldy * 4ull; a.x_bytes = (unsigned long long)rows_in * d->ldx * 4ull;
  a.geo = 1; a.Cin = d->Cin; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.Hi = d->H; a.Wi = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
  hipStream_t st = (hipStream_t)stream;
  p1t_dispatch(a, g, true, st);
  hipLaunchKernelGGL(p1t_reduce_kernel, dim3((d->Cout * N + 63) / 64), dim3(256), 0, st, (const float*)workspace, g.blocks, g.Mtot, g.Ntot, d->Cout, N, dw);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
