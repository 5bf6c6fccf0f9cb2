// The K-class classifier of a segmentation head fused with the BatchNorm + ReLU in front of it (round 6).
//
//   y [rows][C]  --BatchNorm (batch statistics) + ReLU-->  z  --1 x 1 convolution, bias-->  logits [rows][K],   K <= 32, C <= 512
//
// (models/OCR.py:72-74, 97 of the reference: interm_prediction_head[1..4] and conv_bn_dropout[1..2] + conv_out.)  z has ONE consumer, the
// classifier.  The separate passes moved it five times through HBM per training step (535 MB at 8 x 136 x 240 x 512): written by the
// BatchNorm apply, read by the classifier, read again by its backward-weight, and its gradient written by the classifier's backward-data and
// read by both passes of the BatchNorm backward.  Here z and dz exist in registers only:
//   hf_fwd_kernel        logits = relu(bn(y)) Wh^T + bh                                     reads y once
//   hf_bwd_kernel<false> dz = dlogits Wh recomputed per tile; the BatchNorm's sums of (dz masked) and (dz masked) xhat, dWh = dlogits^T z as
//                        per-block slabs, dbh, max|g|, max|y|                                reads y once (+ the 128-byte dlogits rows)
//   hf_bwd_kernel<true>  dz recomputed again; dy = gamma invstd (g - mean(g) - xhat mean(g xhat)) written ONLY as the blocked fp16 x 2
//                        planes of the head layers' kernels (what bn_bwd_apply_h2_kernel writes)  reads y once, writes the planes
// All products are exact fp32 MFMA chains (v_mfma_f32_32x32x2_f32): 2 x 512 x 32 flops per pixel and pass = 55 us of matrix time per pass at
// 8 x 136 x 240 pixels, below the 110 us the pass needs for its 535 MB.  Operand layouts (lane = 32 h + l31):
//   A[i = l31][k = h],  B[k = h][j = l31],  D[i = (r & 3) + 8 (r >> 2) + 4 h][j = l31] in accumulator register r
// and the reduction index may be permuted freely as long as A and B agree -- so a lane's 16 accumulator values (16 rows of one channel) ARE
// its B operands of the backward-weight product over those rows, no transposition.
// Included by norm.hip inside its anonymous namespace.
#ifndef CATSEG_HEADFUSE_H
#define CATSEG_HEADFUSE_H
#include <type_traits>

typedef float hf_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned hf_u32x4 __attribute__((ext_vector_type(4)));
#define HF_ROW(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))

// ---- forward.  Block = kHfFwdWaves waves; a wave owns 32 pixel rows at a time and walks their C channels in chunks of 32 (lane (l31, h): row l31,
// channels c0 + 16 h .. + 15 = 64 contiguous bytes, three chunks of loads in flight under a chunk's 16 MFMAs).  Wh lives in LDS as
// [channel][33] (lane l31 reads class l31 of 16 channels: conflict-free), the BatchNorm constants as [3][C].
constexpr int kHfFwdWaves = 4;      // waves per block: two blocks per CU (67.6 + 6 KB of LDS each); six waves per block measured 207 against 158 us
__global__ __launch_bounds__(kHfFwdWaves * 64, 2) void hf_fwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ mean,
                                                        const float* __restrict__ scale, const float* __restrict__ beta,
                                                        const float* __restrict__ wh, const float* __restrict__ bh, int K, long long rows, int C,
                                                        float* __restrict__ out, int ldo, int zero_to) {
  extern __shared__ __attribute__((aligned(16))) float hf_sm[];
  float* Wl = hf_sm;             // [C][33]
  float* cst = hf_sm + C * 33;   // mean[C], scale[C], beta[C]   (C % 32 == 0: 16-byte aligned)
  for (int k = 0; k < 32; ++k)
    for (int c = threadIdx.x; c < C; c += kHfFwdWaves * 64) Wl[c * 33 + k] = k < K ? wh[(long long)k * C + c] : 0.f;
  for (int c = threadIdx.x; c < C; c += kHfFwdWaves * 64) {
    cst[c] = mean[c];
    cst[C + c] = scale[c];
    cst[2 * C + c] = beta[c];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
  const long long ngroups = (rows + 31) >> 5;
  const float bias = (bh != nullptr && l31 < K) ? bh[l31] : 0.f;
  for (long long g = (long long)blockIdx.x * kHfFwdWaves + wave; g < ngroups; g += (long long)gridDim.x * kHfFwdWaves) {
    const long long row0 = g << 5;
    long long row = row0 + l31;
    if (row >= rows) row = rows - 1;   // (a clamped row's products land in accumulator rows that are not stored)
    const float* yp = y + row * ldy + 16 * h;
    hf_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // ring of four chunk buffers: the loads of chunk c + 3 are issued in front of the MFMAs of chunk c (16 MFMAs = 0.4 us: one chunk of
    // distance did not cover an HBM round trip)
    f32x4 ring[4][4];
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (32 * u < C) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ring[u][q] = ld4(yp + 32 * u + 4 * q);
      }
    for (int c0 = 0; c0 < C; c0 += 128) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cc = c0 + 32 * u;
        if (cc < C) {
          if (cc + 96 < C) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ring[(u + 3) & 3][q] = ld4(yp + cc + 96 + 4 * q);
          }
          const float* cm = cst + cc + 16 * h;
          const float* wl = Wl + (cc + 16 * h) * 33 + l31;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 m4 = *(const f32x4*)(cm + 4 * q), s4 = *(const f32x4*)(cm + C + 4 * q), b4 = *(const f32x4*)(cm + 2 * C + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float z = fmaxf(__builtin_fmaf(ring[u][q][e] - m4[e], s4[e], b4[e]), 0.f);   // = bn_affine + ReLU
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z, wl[(4 * q + e) * 33], acc, 0, 0, 0);
            }
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long ro = row0 + HF_ROW(r, h);
      const float v = acc[r];
      if (ro < rows) {
        if (l31 < K) out[ro * ldo + l31] = v + bias;
        else if (l31 < zero_to) out[ro * ldo + l31] = 0.f;
      }
    }
  }
}

// ---- backward.  Block = NW waves; a wave owns NT 32-channel tiles (block (x, y): channels from 32 NT NW y) whose Wh fragments and BatchNorm
// constants stay in registers, the block walks 32-row chunks blockIdx.x, + gridDim.x, ...  NT = 1, NW = 4: two blocks per CU, the grid sized to be
// resident at once; the y rows and the dl rows of chunk i + 1 load into a second register set from the top of chunk i and are waited for at
// its end.  At 8 x 136 x 240 x 512: second pass 205 us = 5.2 TB/s of (y read + planes written); first pass 247 us (its 32 fp32 MFMAs per
// 32 x 32 tile are 125 us of matrix time; 2 waves per SIMD keep the pipe half busy).
struct HfBwdArgs {
  const float* dl; int lddl;        // gradient of the logits [rows][lddl], lddl >= 32
  const float* y; int ldy;
  const float* stats;               // mean[C], invstd[C]
  const float* gamma; const float* beta;
  const float* wh; int K;           // [K][C]
  long long rows; int C;
  // first pass
  float* part;                      // [gridDim.x][2][C]: sum g, sum g xhat (bn_bwd_finalize_kernel's layout)
  float* dws;                       // [gridDim.x][32][C]: slabs of dWh
  float* dbs;                       // [gridDim.x][32]: slabs of dbh
  unsigned* g_rec; unsigned* y_rec; // amax records: max|g|, max|y|
  // second pass
  const float* coef;                // mean(g)[C], mean(g xhat)[C]
  unsigned char* planes; long long plane_bytes;
  const unsigned* dy_rec; unsigned* scale;
  float* colpart;                   // [gridDim.x][C]: column sums of dy (gradient of a bias in front of the BatchNorm), or null
};

template <bool APPLY, int NT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void hf_bwd_kernel(const HfBwdArgs p) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int cw = ((int)blockIdx.y * NW + wave) * 32 * NT;
  const bool wlive = cw < p.C;
  const int C = p.C, K = p.K, ldy = p.ldy;
  const long long rows = p.rows;
  // the chunk's logits gradient [32 rows][32 classes] (rows past the end and classes past K zeroed), double-buffered: every wave reads it in
  // two layouts (rows on l31 for the dz product, classes on l31 for the dWh product); row stride 36 floats
  __shared__ __attribute__((aligned(16))) float dlS[2][32 * 36];
  // (second pass) a tile of 32 rows x 32 channels leaves through its wave's LDS region as four 1 KB pieces (plane, 16-channel chunk), each
  // contiguous in the blocked planes [2][C / 16][rows][16].  Piece stride 1056 bytes: (plane, chunk, h) map onto eight distinct groups of 8 banks.
  __shared__ __attribute__((aligned(16))) unsigned char hf_stage[APPLY ? NW : 1][APPLY ? 4 * 1056 : 16];
  unsigned gm = 0, ym = 0;
  float mean_[NT], inv_[NT], sc_[NT], be_[NT], c0_[NT], c1_[NT];
  float W[NT][16];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ch = (wlive ? cw : 0) + 32 * t + l31;
    mean_[t] = p.stats[ch];
    inv_[t] = p.stats[C + ch];
    sc_[t] = p.gamma[ch] * inv_[t];      // scale exactly as bn_finalize_kernel stored it
    be_[t] = p.beta[ch];
    c0_[t] = APPLY ? p.coef[ch] : 0.f;
    c1_[t] = APPLY ? p.coef[C + ch] : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) W[t][j] = (16 * h + j < K) ? p.wh[(long long)(16 * h + j) * C + ch] : 0.f;
  }
  float sg[NT], sgx[NT], csum[NT];
  f32x4 dbl4 = {0.f, 0.f, 0.f, 0.f};      // (first pass) the staging thread's share of dbh: its row of every chunk, four classes
  hf_f32x16 accW[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) sg[t] = sgx[t] = csum[t] = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[t][r] = 0.f;
  float sc2 = 1.f;
  if constexpr (APPLY) sc2 = __builtin_ldexpf(1.f, cs_plane_exponent(p.dy_rec[CS_REC_BOUND]));
  const long long nchunks = (rows + 31) >> 5;
  // staging of a chunk's dl: threads 0 .. 255 hold 16 bytes each (row = tid >> 3, classes 4 (tid & 7) ..)
  const int srow = threadIdx.x >> 3, scls = (threadIdx.x & 7) * 4;
  // every vector-memory access of the loop goes through a buffer resource and is issued unconditionally (rows past the end: out of range =
  // zeros read, nothing written): with no branch around a load or a store the compiler's s_waitcnt counts are exact -- the next chunk's rows
  // are waited for with the plane stores still in flight (a conditional store behind them made every wait a vmcnt(0): one full store round
  // trip per chunk)
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dl, (short)0, (int)(unsigned)(rows * p.lddl * 4), 0x00020000);
  const unsigned dl_off = (unsigned)(srow * p.lddl + scls) * 4u, dl_step = 32u * (unsigned)p.lddl * 4u;
  auto fetch_dl = [&](long long ck) {       // (raw: the classes past K are zeroed where the value is written to LDS)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, dl_off + (unsigned)ck * dl_step, 0, 0));
  };
  auto stage_dl = [&](int buf, f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (scls + e < K) ? v[e] : 0.f;
    if constexpr (!APPLY) dbl4 += v;
    if (threadIdx.x < 256) *(f32x4*)(&dlS[buf][srow * 36 + scls]) = v;
  };
  // this lane's 16 rows (accumulator layout) of channel tile t of a chunk: rows (r & 3) + 4 h in four per-lane byte offsets, the chunk and the
  // 8 (r >> 2) rows in a wave-uniform term
  float Y[NT][16];
  const int cbase = wlive ? cw : 0;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + cbase), (short)0, (int)(unsigned)((rows * ldy - cbase) * 4), 0x00020000);
  unsigned yoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) yoff[r] = (unsigned)((r + 4 * h) * ldy + l31) * 4u;
  const unsigned y_step = 32u * (unsigned)ldy * 4u, y_oct = 8u * (unsigned)ldy * 4u;
  auto load_y = [&](float (&Yd)[NT][16], int t, long long ck) {
    const unsigned base = (unsigned)ck * y_step + 128u * t;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Yd[t][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsY, yoff[r & 3] + (base + (r >> 2) * y_oct), 0, 0));
  };
  // per channel: dy = sc g - (k0 + (y - mean) k1) with k0 = sc mean(g), k1 = sc invstd mean(g xhat)  (two FMAs per element)
  float k0_[NT], k1_[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    k0_[t] = sc_[t] * c0_[t];
    k1_[t] = sc_[t] * inv_[t] * c1_[t];
  }
  float gmf = 0.f, ymf = 0.f;
  const unsigned psel = (l31 & 1) ? 0x03020706u : 0x05040100u;      // v_perm selectors of the plane pairing below
  unsigned char* const st = hf_stage[APPLY ? wave : 0];
  unsigned char* const wp = st + (2 * (l31 & 1) + (l31 >> 4)) * 1056 + ((l31 & 15) >> 1) * 4;
  typedef _Float16 hf_h2 __attribute__((ext_vector_type(2)));
  // one resource per (tile, plane, 16-channel chunk) piece of this wave's channels: rows x 32 bytes, a row past the end is out of range
  __amdgpu_buffer_rsrc_t rsP[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      rsP[t][q] = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(APPLY ? p.planes + (q >> 1) * p.plane_bytes + (((long long)(cbase >> 4) + 2 * t + (q & 1)) * rows << 5) : (unsigned char*)nullptr), (short)0,
          (int)(unsigned)((APPLY && wlive) ? rows * 32 : 0), 0x00020000);      // (a wave without channels: empty, its stores are dropped)

  long long ck = blockIdx.x;
  if (ck < nchunks) {
    stage_dl(0, fetch_dl(ck));
#pragma unroll
    for (int t = 0; t < NT; ++t) load_y(Y, t, ck);
  }
  __syncthreads();
  int buf = 0;
  for (; ck < nchunks; ck += gridDim.x) {
    const long long row0 = ck << 5;
    const bool full = row0 + 32 <= rows;
    const long long nk = ck + gridDim.x;
    const bool more = nk < nchunks;
    // the next chunk's dl and y rows start their round trips here, into a second register set; they are waited for at the END of this chunk
    const f32x4 nd = fetch_dl(nk);
    float Yn[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t) load_y(Yn, t, nk);      // (past the last chunk: out of range, zeros)
    __builtin_amdgcn_sched_barrier(0);    // (the round trips start here, not where the scheduler finds room)
    const float* ds = dlS[buf];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // the tile's arithmetic, once for full chunks (no row predicate) and once for the ragged last one
      auto elements = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        hf_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // A of the dz product: dl[row l31][classes 16 h + j]
        const float* ap = ds + l31 * 36 + 16 * h;
        f32x4 a4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[q] = *(const f32x4*)(ap + 4 * q);
        if constexpr (!APPLY) {
          // A of the dWh product: dl[row (j, h)][class l31], all sixteen reads in front of the products (issued one by one inside the chain,
          // each product waited for its own LDS round trip); B = this lane's 16 normalised values, which do not depend on dz -- the two
          // products are independent chains and alternate on the matrix pipe
          float A2[16], zr[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) A2[j] = ds[HF_ROW(j, h) * 36 + l31];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float z = __builtin_fmaf(Y[t][r] - mean_[t], sc_[t], be_[t]);      // = bn_affine: the forward's expression
            zr[r] = FULL ? fmaxf(z, 0.f) : ((z > 0.f && row0 + HF_ROW(r, h) < rows) ? z : 0.f);
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j >> 2][j & 3], W[t][j], acc, 0, 0, 0);
            accW[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[j], zr[j], accW[t], 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float yy = Y[t][r];
            const float a = acc[r];
            const float g = zr[r] > 0.f ? a : 0.f;       // (zr > 0 <=> the element passed the ReLU and its row exists)
            sg[t] += g;
            sgx[t] = __builtin_fmaf(g, (yy - mean_[t]) * inv_[t], sgx[t]);
            gmf = fmaxf(gmf, fabsf(g));
            ymf = fmaxf(ymf, FULL ? fabsf(yy) : (row0 + HF_ROW(r, h) < rows ? fabsf(yy) : 0.f));
          }
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j >> 2][j & 3], W[t][j], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = FULL || row0 + HF_ROW(r, h) < rows;
            const float d = Y[t][r] - mean_[t];
            const float z = __builtin_fmaf(d, sc_[t], be_[t]);      // = bn_affine: the forward's expression
            const bool on = FULL ? (z > 0.f) : ((z > 0.f) && ok);
            const float a = acc[r];
            const float g = on ? a : 0.f;
            float o = __builtin_fmaf(g, sc_[t], -__builtin_fmaf(d, k1_[t], k0_[t]));
            if constexpr (!FULL) o = ok ? o : 0.f;
            csum[t] += o;
            const float xs = o * sc2;
            const _Float16 hh = (_Float16)xs;
            const hf_h2 own2 = {hh, (_Float16)(xs - (float)hh)};
            const unsigned own = __builtin_bit_cast(unsigned, own2);
            // an even lane pairs its high half with its neighbour's and writes the high plane's dword, an odd lane the low plane's
            const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xF, 0xF, false);   // lane ^ 1
            *(unsigned*)(wp + HF_ROW(r, h) * 32) = __builtin_amdgcn_perm(other, own, psel);
          }
        }
      };
      if (full) elements(std::true_type{});
      else elements(std::false_type{});
      if constexpr (APPLY) {       // (no branch around the stores: see rsD)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const hf_u32x4 v = *(const hf_u32x4*)(st + q * 1056 + lane * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rsP[t][q], (unsigned)lane * 16u + (unsigned)row0 * 32u, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __builtin_amdgcn_sched_barrier(0);    // (nothing below moves up into the chunk's work: the copies would drag the loads' wait with them)
    stage_dl(buf ^ 1, nd);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[t][r] = Yn[t][r];
    // the other buffer: every wave finished reading it one barrier ago.  LDS-only barrier (__syncthreads() also drains the vector-memory
    // counter: the plane stores)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    buf ^= 1;
  }
  gm = __float_as_uint(gmf);
  ym = __float_as_uint(ymf);
  // ---- what the block leaves: the two halves of a wave hold different rows of the same channels
  if (wlive) {
    if constexpr (!APPLY) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int ch = cw + 32 * t + l31;
        const float s0 = sg[t] + __shfl_xor(sg[t], 32, 64), s1 = sgx[t] + __shfl_xor(sgx[t], 32, 64);
        if (h == 0) {
          p.part[((long long)blockIdx.x * 2) * C + ch] = s0;
          p.part[((long long)blockIdx.x * 2 + 1) * C + ch] = s1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = accW[t][r];
          p.dws[((long long)blockIdx.x * 32 + HF_ROW(r, h)) * C + ch] = v;
        }
      }
    } else if (p.colpart != nullptr) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float s = csum[t] + __shfl_xor(csum[t], 32, 64);
        if (h == 0) p.colpart[(long long)blockIdx.x * C + cw + 32 * t + l31] = s;
      }
    }
  }
  if constexpr (!APPLY) {
    if (blockIdx.y == 0 && NW == 4) {
      // dbh slab of this block: the 32 staging rows of each class, summed in row order (the staged values past the last chunk are zeros)
      __syncthreads();
      *(f32x4*)(&dlS[0][srow * 36 + scls]) = dbl4;
      __syncthreads();
      if (threadIdx.x < 32) {
        float d = 0.f;
        for (int r = 0; r < 32; ++r) d += dlS[0][r * 36 + threadIdx.x];
        p.dbs[(long long)blockIdx.x * 32 + threadIdx.x] = d;
      }
    }
    if (!wlive) gm = ym = 0;      // (an idle wave computed on wave 0's channels: its sums are dropped, its maxima must be too)
    cs_amax_commit(gm, p.g_rec);
    __syncthreads();              // (cs_amax_commit's LDS words are still being read by thread 0 for the first record)
    cs_amax_commit(ym, p.y_rec);
  } else {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
      const unsigned bound = p.dy_rec[CS_REC_BOUND];
      p.scale[0] = bound;
      ((int*)p.scale)[1] = cs_plane_exponent(bound);
    }
  }
}

// dWh[k][c] = sum over the blocks' slabs, dbh[k] likewise.  Block = 64 outputs x 4 slab groups (slabs b = group mod 4), eight independent
// chains per thread, the groups combined through LDS: every sum in a fixed order
__global__ __launch_bounds__(256) void hf_reduce_kernel(const float* __restrict__ dws, const float* __restrict__ dbs, int nb, int K, int C,
                                                        float* __restrict__ dwh, float* __restrict__ dbh) {
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  const bool isw = i < K * C;
  const int j = i - K * C;                       // (behind the weight: the K bias sums)
  const bool isb = !isw && dbh != nullptr && j < K;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (isw || isb) {
    const int k = isw ? i / C : 0, c = isw ? i - k * C : 0;
    const float* s = isw ? dws + (long long)k * C + c : dbs + j;
    const long long step = isw ? 32ll * C : 32ll;
    int b0 = grp;
    for (; b0 + 28 < nb; b0 += 32) {      // (eight UNCONDITIONAL loads in flight)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s[(long long)(b0 + 4 * u) * step];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += v[u];
    }
    for (; b0 < nb; b0 += 4) a[0] += s[(long long)b0 * step];
  }
  red[grp][o] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (grp == 0) {
    const float v = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    if (isw) dwh[i] = v;
    else if (isb) dbh[j] = v;
  }
}
#endif
