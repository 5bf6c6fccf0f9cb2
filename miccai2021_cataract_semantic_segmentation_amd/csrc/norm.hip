// BatchNorm2d (training statistics, eval affine) fused with ReLU and the residual add,
// forward and backward, on NHWC fp32 activations.  HBM-bound: every pass streams the
// activation once with 16-byte accesses; per-channel reductions are two-level
// (per-block shifted sums -> fp64 Chan merge), deterministic (no atomics).
#include "planes.h"

namespace {

constexpr int kMaxRowBlocks = 1024;

struct RowSplit {
  int tpr;   // threads per row (float4 channel groups handled side by side)
  int rpp;   // rows per pass of a 256-thread block
  int gy;    // channel blocks
  int nrb;   // row blocks
  long long rpb;  // rows per block
};

RowSplit plan_rows(long long rows, int C) {
  RowSplit s;
  const int cpt = (C + 3) / 4;
  s.tpr = cpt < 64 ? cpt : 64;
  s.rpp = 256 / s.tpr;
  s.gy = (cpt + s.tpr - 1) / s.tpr;
  int want = 2048 / s.gy;  // ~2048 blocks in total (8 per CU, 4 independent row loads per thread in flight: the reductions are
  if (want > 1024) want = 1024;   // latency-bound below that); fewer partials = cheaper merge
  if (want < 64) want = 64;
  long long rpb = (rows + want - 1) / want;
  if (rpb < 4 * s.rpp) rpb = 4 * s.rpp;
  rpb = (rpb + s.rpp - 1) / s.rpp * s.rpp;
  s.rpb = rpb;
  s.nrb = (int)((rows + rpb - 1) / rpb);
  return s;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }

// per (row block, channel): shift K, s1 = sum(x-K), s2 = sum((x-K)^2)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ y, int ld, long long rows, int C,
                                                         RowSplit s, float* __restrict__ part) {
  const int t = threadIdx.x;
  const int cg = blockIdx.y * s.tpr + t % s.tpr;
  const int rl = t / s.tpr;
  const int c = cg * 4;
  const long long r0 = (long long)blockIdx.x * s.rpb;
  const long long r1 = min(r0 + s.rpb, rows);
  const bool act = (c < C) && (rl < s.rpp);
  f32x4 K = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  if (act) {
    K = ld4(y + r0 * ld + c);
    const long long st = s.rpp;
    long long r = r0 + rl;
    for (; r + 3 * st < r1; r += 4 * st) {   // four independent loads in flight
      const f32x4 d0 = ld4(y + r * ld + c) - K, d1 = ld4(y + (r + st) * ld + c) - K;
      const f32x4 d2 = ld4(y + (r + 2 * st) * ld + c) - K, d3 = ld4(y + (r + 3 * st) * ld + c) - K;
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    for (; r < r1; r += st) {
      const f32x4 d = ld4(y + r * ld + c) - K;
      s1 += d;
      s2 += d * d;
    }
  }
  __shared__ f32x4 sh1[256], sh2[256];
  sh1[t] = s1;
  sh2[t] = s2;
  __syncthreads();
  if (act && rl == 0) {
    for (int k = 1; k < s.rpp; ++k) {
      s1 += sh1[t + k * s.tpr];
      s2 += sh2[t + k * s.tpr];
    }
    float* o = part + ((long long)blockIdx.x * 3) * C;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c + i < C) {
        o[c + i] = K[i];
        o[C + c + i] = s1[i];
        o[2 * C + c + i] = s2[i];
      }
  }
}

// merge of the per-row-block partials (K, s1, s2): block = 16 channels x 64 row-block lanes, ONE pass over the L2-resident
// partials with 4 independent load chains per thread (the convolution epilogues produce up to M / 64 = 4080 of them per layer:
// the two-pass form of round 1 spent 19 us per launch on dependent loads), fp64 throughout:
// Block = 4 channels x 256 row-block lanes (<= 16 partials per thread in 4 independent chains: ~4 L2 round trips).
//   mean_b = K_b + s1_b / n_b,   M2_b = s2_b - s1_b^2 / n_b
//   mean = sum_b n_b mean_b / n,   M2 = sum_b (M2_b + n_b mean_b^2) - n mean^2
// (the last subtraction cancels at most mean^2 / var digits of the 16 fp64 carries: exact to fp32 for |mean| / std < 1e4)
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int nrb, long long rpb, long long rows, int C,
                                                           const float* __restrict__ gamma, float eps, float momentum, float* running_mean,
                                                           float* running_var, float* __restrict__ stats, float* __restrict__ scale,
                                                           const int* __restrict__ counts, const unsigned* __restrict__ y_rec = nullptr,
                                                           unsigned* __restrict__ z_rec = nullptr, const float* __restrict__ beta = nullptr) {
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  const bool live = c < C;
  const double n_full = (double)rpb, inv_full = 1.0 / n_full;
  const double n_last = (double)(rows - (long long)(nrb - 1) * rpb), inv_last = 1.0 / n_last;
  __shared__ double sh0[16][5], sh1[16][5];
  double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
  // what the channel's LAST lane needs after the merge is fetched now, in the shadow of the partial loads (these launches are a chain of
  // dependent memory round trips; every one taken off the tail is ~1 us of 10)
  float gam = 0.f, bet = 0.f, rm = 0.f, rv = 0.f;
  unsigned ym = 0;
  if (live && rl == 0) {
    gam = gamma[c];
    if (running_mean) { rm = running_mean[c]; rv = running_var[c]; }
    if (z_rec) {
      bet = beta[c];
      for (int i = 0; i < CS_AMAX_SLOTS; ++i) ym = max(ym, y_rec[i * CS_AMAX_STRIDE]);
    }
  }
  if (live) {
    // four independent load chains per round.  Every load is UNCONDITIONAL (a chain past the end re-reads the last row and is discarded by a
    // select): with a branch around each chain's loads the compiler waited for one chain before it issued the next -- four dependent L2 round
    // trips per round (the disassembly showed load, s_waitcnt vmcnt(0), load, ...), most of these launches' 8 us (round 6)
    for (int b = rl; b < nrb; b += 1024) {
      int cnt[4];
      float K[4], t1[4], t2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int bb = min(b + 256 * u, nrb - 1);
        const float* o = part + ((long long)bb * 3) * C;
        // the count and the partial row are fetched TOGETHER (the row of an empty tile -- a wave of csrc/dconv3_pl.hip whose pixel rows all
        // lie below the image -- is allocated but never written: whatever it holds is discarded by the selects below, never multiplied in)
        cnt[u] = counts ? counts[bb] : 1;   // per-block row counts (2-D pixel tiles with ragged edges: csrc/dconv3_b3.hip)
        K[u] = o[c]; t1[u] = o[C + c]; t2[u] = o[2 * C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int bb = b + 256 * u;
        const bool last = bb == nrb - 1;
        double nb = last ? n_last : n_full, inv = last ? inv_last : inv_full;
        if (counts) {
          nb = (double)cnt[u];
          inv = 1.0 / (cnt[u] > 0 ? nb : 1.0);
        }
        const double dK = K[u], d1 = t1[u], d2 = t2[u];
        const double mb = dK + d1 * inv;
        const double u0 = nb * mb, u1 = (d2 - d1 * d1 * inv) + nb * mb * mb;
        const bool use = bb < nrb && cnt[u] > 0;
        a0[u] += use ? u0 : 0.0;
        a1[u] += use ? u1 : 0.0;
      }
    }
  }
  // merge over the 256 row lanes in a fixed order (deterministic): the 16 row lanes of a wave by lane shuffles (lane = 4 row lane + channel),
  // the 16 waves through LDS -- two block barriers instead of the nine of a shared-memory tree (these launches are pure latency: 628 per step)
  double r0 = (a0[0] + a0[1]) + (a0[2] + a0[3]), r1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
#pragma unroll
  for (int o = 4; o <= 32; o <<= 1) {
    r0 += __shfl_xor(r0, o, 64);
    r1 += __shfl_xor(r1, o, 64);
  }
  if ((threadIdx.x & 63) < 4) {
    sh0[threadIdx.x >> 6][cl] = r0;
    sh1[threadIdx.x >> 6][cl] = r1;
  }
  __syncthreads();
  if (rl != 0 || !live) return;
  double t0 = 0, t1 = 0;
  for (int w = 0; w < 16; ++w) {
    t0 += sh0[w][cl];
    t1 += sh1[w][cl];
  }
  const double mean = t0 / (double)rows;
  double m2 = t1 - (double)rows * mean * mean;
  if (m2 < 0) m2 = 0;
  const double n = (double)rows;
  const float var = (float)(m2 / n);
  const float invstd = 1.0f / sqrtf(var + eps);
  stats[c] = (float)mean;
  stats[C + c] = invstd;
  scale[c] = gam * invstd;
  if (z_rec) {
    // bound of the normalised output of this channel, BEFORE the pass that computes it: |fma(y - mean, scale, beta)| <= |scale| (max|y| +
    // |mean|) + |beta|, with max|y| from the convolution's epilogue (csrc/dconv3_pl.hip).  The maximum over the channels (positive floats
    // order like their bit patterns) is what catseg_bn_apply_planes derives the planes' exponent from (csrc/planes.h).
    const float bound = (fabsf(gam * invstd) * (__uint_as_float(ym) + fabsf((float)mean)) + fabsf(bet)) * 1.001f;
    atomicMax(z_rec + CS_REC_BOUND, __float_as_uint(bound));
  }
  if (running_mean) {
    const float unb = (float)(m2 / (n > 1 ? n - 1 : 1));
    running_mean[c] = (1.f - momentum) * rm + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * rv + momentum * unb;
  }
}

__global__ void bn_eval_kernel(int C, const float* gamma, const float* rv, float eps, float* scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) scale[c] = gamma[c] / sqrtf(rv[c] + eps);
}

// v = fma(y - mean, scale, beta), spelled out so that forward and backward evaluate the identical expression: the backward
// recomputes the ReLU mask from y instead of reading z when there is no residual branch (one tensor read less)
__device__ __forceinline__ f32x4 bn_affine(f32x4 y, f32x4 mean, f32x4 scale, f32x4 beta) {
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(y[i] - mean[i], scale[i], beta[i]);
  return v;
}

// z = act((y - mean) * scale + beta (+ res))
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ mean,
                                                       const float* __restrict__ scale, const float* __restrict__ beta,
                                                       const float* __restrict__ res, int ldr, float* __restrict__ z,
                                                       int ldz, long long rows, int C, int relu, unsigned* __restrict__ amax,
                                                       unsigned char* __restrict__ mask = nullptr) {
  const int cpt = C >> 2;
  unsigned m = 0;
  CS_QUAD_LOOP(rows, cpt, r, c) {
    f32x4 v = bn_affine(ld4(y + r * ldy + c), ld4(mean + c), ld4(scale + c), ld4(beta + c));
    if (res) v += ld4(res + r * ldr + c);
    if (relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    *(f32x4*)(z + r * ldz + c) = v;
    m = max(m, cs_abs_bits4(v));
    if (mask) {
      // the ReLU mask as BITS: bit (e & 7) of byte e >> 3, e = r C + c the flat element index (C % 8 == 0).  This thread holds a nibble; its
      // neighbour lane ^ 1 holds the other half of the byte (consecutive lanes walk consecutive quads, an even number per row and per stride:
      // both lanes of a pair are in the loop together).  The backward passes read 1 bit per element instead of z.
      const unsigned nib = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
      const unsigned other = (unsigned)__shfl_xor((int)nib, 1, 64);
      if ((c & 4) == 0) mask[(r * C + c) >> 3] = (unsigned char)(nib | (other << 4));
    }
  }
  if (amax) cs_amax_commit(m, amax);       // (uniform per launch: the shuffles run with every lane)
}

// The same pass writing the fp16 x 2 operand planes of z (csrc/planes.h) -- and z itself only if someone reads it as fp32 (zf != nullptr:
// the residual branch of the next block, the HRNet fuse layers; the first convolution's output inside a BasicBlock has one consumer, the
// direct 3x3 kernel, and exists as planes only).  Tiles of 64 rows x 64 channels per block iteration; the exponent comes from the bound
// bn_finalize_kernel left in the record (+ max|residual| from the residual's own record).
__global__ __launch_bounds__(256) void bn_apply_planes_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ mean,
                                                              const float* __restrict__ scale, const float* __restrict__ beta,
                                                              const float* __restrict__ res, int ldr, const unsigned* __restrict__ res_rec,
                                                              float* __restrict__ zf, int ldz, unsigned char* __restrict__ planes,
                                                              long long rows, int C, int relu, unsigned* __restrict__ rec,
                                                              unsigned char* __restrict__ mask = nullptr) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[CsPlaneTile::BYTES];
  float bound = __uint_as_float(rec[CS_REC_BOUND]);
  if (res) bound += __uint_as_float(cs_amax_read(res_rec));
  const int e = cs_plane_exponent(__float_as_uint(bound));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ((int*)rec)[CS_REC_EXP] = e;
    rec[CS_REC_FINAL] = __float_as_uint(bound);
  }
  const float sc = __builtin_ldexpf(1.f, e);
  const int NG = C >> 3, gblocks = (NG + 7) >> 3;
  const long long rblocks = (rows + CsPlaneTile::ROWS - 1) / CsPlaneTile::ROWS;
  unsigned m = 0;
  for (long long t = blockIdx.x; t < rblocks * gblocks; t += gridDim.x) {
    const long long row0 = (t / gblocks) * CsPlaneTile::ROWS;
    const int g0 = (int)(t % gblocks) << 3, ng = NG - g0 < 8 ? NG - g0 : 8;
    for (int i = threadIdx.x; i < CsPlaneTile::ROWS * ng; i += 256) {
      const int row = i / ng, g = i - row * ng;
      if (row0 + row < rows) {
        const long long r = row0 + row;
        const int c = (g0 + g) * 8;
        f32x4 v0 = bn_affine(ld4(y + r * ldy + c), ld4(mean + c), ld4(scale + c), ld4(beta + c));
        f32x4 v1 = bn_affine(ld4(y + r * ldy + c + 4), ld4(mean + c + 4), ld4(scale + c + 4), ld4(beta + c + 4));
        if (res) {
          v0 += ld4(res + r * ldr + c);
          v1 += ld4(res + r * ldr + c + 4);
        }
        if (relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
        }
        if (zf) {
          *(f32x4*)(zf + r * ldz + c) = v0;
          *(f32x4*)(zf + r * ldz + c + 4) = v1;
        }
        if (mask) {       // the ReLU mask as bits (bn_apply_kernel): this thread's 8 channels are one byte
          unsigned b = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) b |= (v0[k] > 0.f ? 1u << k : 0u) | (v1[k] > 0.f ? 16u << k : 0u);
          mask[(r * C + c) >> 3] = (unsigned char)b;
        }
        m = max(m, max(cs_abs_bits4(v0), cs_abs_bits4(v1)));
        const float xs[8] = {v0[0] * sc, v0[1] * sc, v0[2] * sc, v0[3] * sc, v1[0] * sc, v1[1] * sc, v1[2] * sc, v1[3] * sc};
        CsPlaneTile::stage(sm, row, g, xs);
      }
    }
    __syncthreads();
    const long long left = rows - row0;
    CsPlaneTile::flush(sm, planes, rows, NG, row0, left < CsPlaneTile::ROWS ? (int)left : CsPlaneTile::ROWS, g0, ng);
    __syncthreads();
  }
  cs_amax_commit(m, rec);      // max|z| as it turned out (the next layer's bound adds it when z is its residual)
}

// backward partials: sg = sum g, sgx = sum g * xhat
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ z,
                                                             int ldz, const float* __restrict__ y, int ldy,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, long long rows, int C, int relu,
                                                             RowSplit s, float* __restrict__ part, unsigned* __restrict__ gmax_rec = nullptr,
                                                             unsigned* __restrict__ ymax_rec = nullptr,
                                                             const unsigned char* __restrict__ mask = nullptr) {
  const int t = threadIdx.x;
  const int cg = blockIdx.y * s.tpr + t % s.tpr;
  const int rl = t / s.tpr;
  const int c = cg * 4;
  const long long r0 = (long long)blockIdx.x * s.rpb;
  const long long r1 = min(r0 + s.rpb, rows);
  const bool act = (c < C) && (rl < s.rpp);
  f32x4 sg = {0, 0, 0, 0}, sgx = {0, 0, 0, 0};
  unsigned gm = 0, ym = 0;
  if (act) {
    const f32x4 mean = ld4(stats + c), inv = ld4(stats + C + c);
    f32x4 sc = {0, 0, 0, 0}, be = {0, 0, 0, 0};
    if (relu && !z && !mask) { sc = ld4(gamma + c) * inv; be = ld4(beta + c); }   // scale exactly as bn_finalize stored it
    const long long st = s.rpp;
    long long r = r0 + rl;
    for (; r + 3 * st < r1; r += 4 * st) {   // four independent rows in flight (8 - 12 loads)
      f32x4 g[4], yy[4], zz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = ld4(dz + (r + u * st) * lddz + c);
        yy[u] = ld4(y + (r + u * st) * ldy + c);
      }
      if (relu && mask) {        // (this quad's nibble of the mask byte (r C + c) >> 3)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned nib = (unsigned)mask[((r + u * st) * C + c) >> 3] >> (c & 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) g[u][i] = ((nib >> i) & 1u) ? g[u][i] : 0.f;
        }
      } else if (relu) {
#pragma unroll
        for (int u = 0; u < 4; ++u) zz[u] = z ? ld4(z + (r + u * st) * ldz + c) : bn_affine(yy[u], mean, sc, be);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) g[u][i] = zz[u][i] > 0.f ? g[u][i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 xh = (yy[u] - mean) * inv;
        sg += g[u];
        sgx += g[u] * xh;
        gm = max(gm, cs_abs_bits4(g[u]));
        ym = max(ym, cs_abs_bits4(yy[u]));
      }
    }
    for (; r < r1; r += st) {
      f32x4 g = ld4(dz + r * lddz + c);
      const f32x4 yy = ld4(y + r * ldy + c);
      if (relu && mask) {
        const unsigned nib = (unsigned)mask[(r * C + c) >> 3] >> (c & 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = ((nib >> i) & 1u) ? g[i] : 0.f;
      } else if (relu) {
        const f32x4 zz = z ? ld4(z + r * ldz + c) : bn_affine(yy, mean, sc, be);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = zz[i] > 0.f ? g[i] : 0.f;
      }
      const f32x4 xh = (yy - mean) * inv;
      sg += g;
      sgx += g * xh;
      gm = max(gm, cs_abs_bits4(g));
      ym = max(ym, cs_abs_bits4(yy));
    }
  }
  __shared__ f32x4 sh1[256], sh2[256];
  sh1[t] = sg;
  sh2[t] = sgx;
  __syncthreads();
  if (act && rl == 0) {
    for (int k = 1; k < s.rpp; ++k) {
      sg += sh1[t + k * s.tpr];
      sgx += sh2[t + k * s.tpr];
    }
    float* o = part + ((long long)blockIdx.x * 2) * C;
    *(f32x4*)(o + c) = sg;
    *(f32x4*)(o + C + c) = sgx;
  }
  if (gmax_rec) cs_amax_commit(gm, gmax_rec);     // max |masked gradient|: the bound of the output's planes needs it (bn_bwd_finalize_kernel)
  if (ymax_rec) {                                 // max |y|, for a layer whose forward left none (the head layers: bn_bwd_apply_h2_kernel)
    __syncthreads();                              // (cs_amax_commit's LDS words are still being read by thread 0 for the first record)
    cs_amax_commit(ym, ymax_rec);
  }
}

// 4 channels x 256 row lanes per block (the geometry of bn_finalize_kernel): up to 2040 partial rows (one per pixel tile of the direct
// kernels) are two rounds of four INDEPENDENT loads per thread (with 16 channels x 64 row lanes they were 32 deep, four in flight: a chain
// of ~8 L2 round trips on 3 - 24 blocks); fixed summation order
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nrb, long long rows, int C,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef,
                                                               const unsigned* __restrict__ gmax_rec = nullptr,
                                                               const unsigned* __restrict__ y_rec = nullptr, const float* __restrict__ stats = nullptr,
                                                               const float* __restrict__ gamma = nullptr, unsigned* __restrict__ dy_rec = nullptr) {
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  const bool live = c < C;
  // (the tail's inputs fetched in the shadow of the partial loads, as in bn_finalize_kernel)
  unsigned gm = 0, ym = 0;
  float st_mean = 0.f, st_inv = 0.f, gam = 0.f;
  if (dy_rec && rl == 0 && live) {
    for (int i = 0; i < CS_AMAX_SLOTS; ++i) {
      gm = max(gm, gmax_rec[i * CS_AMAX_STRIDE]);
      ym = max(ym, y_rec[i * CS_AMAX_STRIDE]);
    }
    st_mean = stats[c]; st_inv = stats[C + c]; gam = gamma[c];
  }
  double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
  if (live)
    for (int b = rl; b < nrb; b += 1024) {
      // UNCONDITIONAL loads (a chain past the end re-reads the last row, a select discards it): see bn_finalize_kernel
      float v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* o = part + ((long long)min(b + 256 * u, nrb - 1) * 2) * C;
        v0[u] = o[c];
        v1[u] = o[C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool use = b + 256 * u < nrb;
        a0[u] += use ? (double)v0[u] : 0.0;
        a1[u] += use ? (double)v1[u] : 0.0;
      }
    }
  double sg = (a0[0] + a0[1]) + (a0[2] + a0[3]), sgx = (a1[0] + a1[1]) + (a1[2] + a1[3]);
  // the 16 row lanes of a wave by lane shuffles (lane = 4 row lane + channel), the 16 waves through LDS (fixed order)
#pragma unroll
  for (int o = 4; o <= 32; o <<= 1) {
    sg += __shfl_xor(sg, o, 64);
    sgx += __shfl_xor(sgx, o, 64);
  }
  __shared__ double s1[16][5], s2[16][5];
  if ((threadIdx.x & 63) < 4) { s1[threadIdx.x >> 6][cl] = sg; s2[threadIdx.x >> 6][cl] = sgx; }
  __syncthreads();
  if (rl != 0 || !live) return;
  sg = 0; sgx = 0;
  for (int k = 0; k < 16; ++k) { sg += s1[k][cl]; sgx += s2[k][cl]; }
  if (dbeta) dbeta[c] = (float)sg;
  if (dgamma) dgamma[c] = (float)sgx;
  const float mg = (float)(sg / (double)rows), mgx = (float)(sgx / (double)rows);
  coef[c] = mg;
  coef[C + c] = mgx;
  if (dy_rec) {
    // bound of dy = gamma invstd (g - mean(g) - xhat mean(g xhat)) for this channel, before the pass that computes it:
    //   |dy| <= |gamma invstd| (max|g| + |mean(g)| + max|xhat| |mean(g xhat)|),   max|xhat| <= (max|y| + |mean|) invstd
    const float inv = st_inv, xh = (__uint_as_float(ym) + fabsf(st_mean)) * inv;
    const float bound = fabsf(gam * inv) * (__uint_as_float(gm) + fabsf(mg) + xh * fabsf(mgx)) * 1.001f;
    atomicMax(dy_rec + CS_REC_BOUND, __float_as_uint(bound));
  }
}

// bn_bwd_apply_kernel writing dy as fp16 x 2 planes ONLY (csrc/planes.h): its two consumers -- backward-data and backward-weight of the
// convolution in front of this BatchNorm, csrc/dconv3_pl.hip / dwgrad3_pl.hip -- stream planes.  The residual gradient stays fp32.
__global__ __launch_bounds__(256) void bn_bwd_apply_planes_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ z, int ldz,
                                                                  const float* __restrict__ y, int ldy, const float* __restrict__ stats,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ coef, long long rows, int C, int relu,
                                                                  unsigned char* __restrict__ planes, unsigned* __restrict__ rec,
                                                                  float* __restrict__ dres, int lddres, int dres_acc,
                                                                  const unsigned char* __restrict__ mask = nullptr) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[CsPlaneTile::BYTES];
  const int e = cs_plane_exponent(rec[CS_REC_BOUND]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ((int*)rec)[CS_REC_EXP] = e;
    rec[CS_REC_FINAL] = rec[CS_REC_BOUND];
  }
  const float sc = __builtin_ldexpf(1.f, e);
  const int NG = C >> 3, gblocks = (NG + 7) >> 3;
  const long long rblocks = (rows + CsPlaneTile::ROWS - 1) / CsPlaneTile::ROWS;
  for (long long t = blockIdx.x; t < rblocks * gblocks; t += gridDim.x) {
    const long long row0 = (t / gblocks) * CsPlaneTile::ROWS;
    const int g0 = (int)(t % gblocks) << 3, ng = NG - g0 < 8 ? NG - g0 : 8;
    for (int i = threadIdx.x; i < CsPlaneTile::ROWS * ng; i += 256) {
      const int row = i / ng, g8 = i - row * ng;
      if (row0 + row < rows) {
        const long long r = row0 + row;
        float xs[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = (g0 + g8) * 8 + 4 * q;
          f32x4 g = ld4(dz + r * lddz + c);
          const f32x4 inv = ld4(stats + C + c), mean = ld4(stats + c);
          const f32x4 yy = ld4(y + r * ldy + c);
          if (relu && mask) {
            const unsigned nib = (unsigned)mask[(r * C + c) >> 3] >> (c & 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = ((nib >> k) & 1u) ? g[k] : 0.f;
          } else if (relu) {
            const f32x4 zz = z ? ld4(z + r * ldz + c) : bn_affine(yy, mean, ld4(gamma + c) * inv, ld4(beta + c));
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = zz[k] > 0.f ? g[k] : 0.f;
          }
          const f32x4 xh = (yy - mean) * inv;
          const f32x4 o = ld4(gamma + c) * inv * (g - ld4(coef + c) - xh * ld4(coef + C + c));     // = bn_bwd_apply_kernel's expression
#pragma unroll
          for (int k = 0; k < 4; ++k) xs[4 * q + k] = o[k] * sc;
          if (dres) {
            f32x4* d = (f32x4*)(dres + r * lddres + c);
            *d = dres_acc ? (*d + g) : g;
          }
        }
        CsPlaneTile::stage(sm, row, g8, xs);
      }
    }
    __syncthreads();
    const long long left = rows - row0;
    CsPlaneTile::flush(sm, planes, rows, NG, row0, left < CsPlaneTile::ROWS ? (int)left : CsPlaneTile::ROWS, g0, ng);
    __syncthreads();
  }
}

// bn_bwd_apply_kernel writing dy ONLY as the BLOCKED fp16 x 2 planes of the head layers' kernels (csrc/igemm_f16x2.hip: [2][C / 16][rows][16],
// what catseg_split2h produces from an fp32 dy) -- the fp32 tensor, the split pass that read it back and the bias gradient's pass over it
// disappear (round 5; per 512-channel head layer at 8 x 136 x 240: 535 MB written + 2 x 535 MB read).  The exponent comes from the bound
// bn_bwd_finalize_kernel left in dy_rec, as for the trunk's planes; scale = the 8-byte record {bits of the bound, e} the consuming kernels read.
// Grid (row blocks, channel blocks of 64): a block walks the 128-row tiles bx, bx + gridDim.x, ... of ITS 64 channels, so that the column
// sums of dy -- the gradient of a convolution bias in front of the BatchNorm (models/OCR.py:72-74; mathematically 0, rounding noise in any
// implementation) -- accumulate in registers: colpart[bx][C], merged by colsum_rows_kernel in a fixed order.  Requires C % 64 == 0.
__global__ __launch_bounds__(256) void bn_bwd_apply_h2_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ z, int ldz,
                                                              const float* __restrict__ y, int ldy, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ coef, long long rows, int C, int relu,
                                                              unsigned char* __restrict__ planes, long long plane_bytes,
                                                              const unsigned* __restrict__ dy_rec, unsigned* __restrict__ scale,
                                                              float* __restrict__ colpart) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[CsPlaneTile::BYTES];
  const unsigned bound = dy_rec[CS_REC_BOUND];
  const int e = cs_plane_exponent(bound);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    scale[0] = bound;
    ((int*)scale)[1] = e;
  }
  const float sc = __builtin_ldexpf(1.f, e);
  const int g0 = blockIdx.y * 8, g8 = threadIdx.x & 7, c = (g0 + g8) * 8;     // this thread's 8 channels, for every tile
  const long long rblocks = (rows + CsPlaneTile::ROWS - 1) / CsPlaneTile::ROWS;
  f32x4 ga[2], ia[2], me[2], be[2], c0[2], c1[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    ia[q] = ld4(stats + C + c + 4 * q); me[q] = ld4(stats + c + 4 * q);
    ga[q] = ld4(gamma + c + 4 * q);
    be[q] = beta ? ld4(beta + c + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    c0[q] = ld4(coef + c + 4 * q); c1[q] = ld4(coef + C + c + 4 * q);
  }
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long rt = blockIdx.x; rt < rblocks; rt += gridDim.x) {
    const long long row0 = rt * CsPlaneTile::ROWS;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = (threadIdx.x >> 3) + 32 * k;
      if (row0 + row < rows) {
        const long long r = row0 + row;
        float xs[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f32x4 g = ld4(dz + r * lddz + c + 4 * q);
          const f32x4 yy = ld4(y + r * ldy + c + 4 * q);
          if (relu) {
            const f32x4 zz = z ? ld4(z + r * ldz + c + 4 * q) : bn_affine(yy, me[q], ga[q] * ia[q], be[q]);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = zz[j] > 0.f ? g[j] : 0.f;
          }
          const f32x4 xh = (yy - me[q]) * ia[q];
          const f32x4 o = ga[q] * ia[q] * (g - c0[q] - xh * c1[q]);     // = bn_bwd_apply_kernel's expression
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            xs[4 * q + j] = o[j] * sc;
            cs[4 * q + j] += o[j];
          }
        }
        CsPlaneTile::stage(sm, row, g8, xs);
      }
    }
    __syncthreads();
    // the tile leaves as 32-byte (row, 16-channel chunk) pieces, consecutive rows of a chunk back to back: 2 KB runs per wave instruction
    const long long left = rows - row0;
    const int nrows = left < CsPlaneTile::ROWS ? (int)left : CsPlaneTile::ROWS;
    for (int i = threadIdx.x; i < 2 * CsPlaneTile::GROUPS * CsPlaneTile::ROWS; i += 256) {
      const int half = i & 1, row = (i >> 1) & (CsPlaneTile::ROWS - 1), cp = (i >> 8) & 3, p = i >> 10;
      if (row < nrows)
        *(cs_h8*)(planes + p * plane_bytes + (((long long)((g0 >> 1) + cp) * rows + row0 + row) << 5) + half * 16) =
            *(const cs_h8*)(sm + p * CsPlaneTile::PS + (2 * cp + half) * CsPlaneTile::GS + row * 16);
    }
    __syncthreads();
  }
  if (colpart) {
    float* red = (float*)sm;          // [32 row lanes][64 channels + 1]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(threadIdx.x >> 3) * 65 + g8 * 8 + j] = cs[j];
    __syncthreads();
    if (threadIdx.x < 64) {
      float t = 0.f;
      for (int k = 0; k < 32; ++k) t += red[k * 65 + threadIdx.x];
      colpart[(long long)blockIdx.x * C + g0 * 8 + threadIdx.x] = t;
    }
  }
}

// out[c] = sum over the partial rows part[0 .. nparts)[c] in a fixed order: block = 64 channels x 4 row groups (group g takes rows g, g + 4, ...),
// eight independent chains per thread, the groups merged through LDS (256 rows were 64 dependent rounds of four loads: 30 us)
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int t = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + t;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    int i0 = grp;
    for (; i0 + 28 < nparts; i0 += 32) {      // (eight UNCONDITIONAL loads in flight)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long long)(i0 + 4 * u) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += v[u];
    }
    for (; i0 < nparts; i0 += 4) a[0] += part[(long long)i0 * C + c];
  }
  red[grp][t] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (grp == 0 && c < C) out[c] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ z,
                                                           int ldz, const float* __restrict__ y, int ldy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ coef, long long rows, int C,
                                                           int relu, float* __restrict__ dy, int lddy, float* __restrict__ dres, int lddres,
                                                           int dres_acc, unsigned* __restrict__ amax,
                                                           const unsigned char* __restrict__ mask = nullptr) {
  const int cpt = C >> 2;
  unsigned m = 0;
  CS_QUAD_LOOP(rows, cpt, r, c) {
    f32x4 g = ld4(dz + r * lddz + c);
    const f32x4 inv = ld4(stats + C + c), mean = ld4(stats + c);
    const f32x4 yy = ld4(y + r * ldy + c);
    if (relu && mask) {
      const unsigned nib = (unsigned)mask[(r * C + c) >> 3] >> (c & 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = ((nib >> k) & 1u) ? g[k] : 0.f;
    } else if (relu) {
      const f32x4 zz = z ? ld4(z + r * ldz + c) : bn_affine(yy, mean, ld4(gamma + c) * inv, ld4(beta + c));
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = zz[k] > 0.f ? g[k] : 0.f;
    }
    const f32x4 xh = (yy - mean) * inv;
    const f32x4 o = ld4(gamma + c) * inv * (g - ld4(coef + c) - xh * ld4(coef + C + c));
    *(f32x4*)(dy + r * lddy + c) = o;
    m = max(m, cs_abs_bits4(o));
    if (dres) {
      f32x4* d = (f32x4*)(dres + r * lddres + c);
      *d = dres_acc ? (*d + g) : g;
    }
  }
  if (amax) cs_amax_commit(m, amax);
}

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

#include "headfuse.h"

}  // namespace

// eval-mode BatchNorm folded into the preceding convolution: w'[o,:] = w[o,:] * g[o]/sqrt(rv[o]+eps),
// b'[o] = beta[o] + (b[o] - rm[o]) * g[o]/sqrt(rv[o]+eps)
__global__ void fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                               int O, int per_out, float* __restrict__ wf, float* __restrict__ bf) {
  const int o = blockIdx.x;
  // the scale is formed in fp64 so that each folded weight / bias carries ONE fp32 rounding (a per-channel error of the
  // scale would be systematic over the whole reduction)
  const double sc = (double)gamma[o] / sqrt((double)rv[o] + (double)eps);
  for (int i = threadIdx.x; i < per_out; i += blockDim.x)
    wf[(long long)o * per_out + i] = (float)((double)w[(long long)o * per_out + i] * sc);
  if (threadIdx.x == 0) bf[o] = (float)((double)beta[o] + ((double)(b ? b[o] : 0.f) - (double)rm[o]) * sc);
}
extern "C" int catseg_fold_bn(const float* w, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                              const float* running_var, float eps, int O, int per_out, float* w_folded, float* bias_folded,
                              catseg_stream_t stream) {
  CS_REQUIRE(O > 0 && per_out > 0, "fold_bn: bad args");
  hipLaunchKernelGGL(fold_bn_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, w, bias, gamma, beta, running_mean, running_var, eps, O,
                     per_out, w_folded, bias_folded);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" size_t catseg_bn_workspace(long long rows, int C) {
  (void)rows;
  return cs_align_up((size_t)(kMaxRowBlocks * 3 + 2) * (size_t)((C + 3) & ~3) * 4, 256);
}

extern "C" int catseg_bn_train_stats(const float* y, long long rows, int C, int ldy, const float* gamma, float eps,
                                     float momentum, float* running_mean, float* running_var, float* stats_out,
                                     float* scale, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0 && ldy >= C, "bn stats: C and ld must be multiples of 4");
  CS_REQUIRE(cs_aligned16(y) && cs_aligned16(stats_out) && cs_aligned16(scale), "bn stats: alignment");
  if (workspace_bytes < catseg_bn_workspace(rows, C) || !workspace) {
    catseg_set_error("bn stats: workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const RowSplit s = plan_rows(rows, C);
  float* part = (float*)workspace;
  hipLaunchKernelGGL(bn_partial_kernel, dim3(s.nrb, s.gy), dim3(256), 0, st, y, ldy, rows, C, s, part);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, (const float*)part, s.nrb, s.rpb, rows, C,
                     gamma, eps, momentum, running_mean, running_var, stats_out, scale, (const int*)nullptr);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// the second half of catseg_bn_train_stats on its own: merges [n_blocks][3][C] partials (K, s1, s2) over row blocks of
// rows_per_block rows (the last one shorter) -- written by catseg_conv2d_fwd_bnstats / _bf16x3_bnstats
extern "C" int catseg_bn_finalize(const float* partials, int n_blocks, long long rows_per_block, long long rows, int C, const float* gamma,
                                  float eps, float momentum, float* running_mean, float* running_var, float* stats_out, float* scale,
                                  catseg_stream_t stream) {
  CS_REQUIRE(partials && n_blocks > 0 && rows > 0 && C > 0 && (long long)(n_blocks - 1) * rows_per_block < rows &&
                 (long long)n_blocks * rows_per_block >= rows, "bn finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, (hipStream_t)stream, partials, n_blocks, rows_per_block, rows, C,
                     gamma, eps, momentum, running_mean, running_var, stats_out, scale, (const int*)nullptr);
#ifdef BN_AB_DOUBLE_FINALIZE
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, (hipStream_t)stream, partials, n_blocks, rows_per_block, rows, C,
                     gamma, eps, momentum, running_mean, running_var, stats_out, scale, (const int*)nullptr);
#endif
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// the same merge over blocks with individual row counts (counts[b] >= 1, sum = rows): the 2-D pixel tiles of catseg_dconv3
extern "C" int catseg_bn_finalize_counts(const float* partials, int n_blocks, const int* counts, long long rows, int C, const float* gamma,
                                         float eps, float momentum, float* running_mean, float* running_var, float* stats_out, float* scale,
                                         catseg_stream_t stream) {
  CS_REQUIRE(partials && counts && n_blocks > 0 && rows > 0 && C > 0, "bn finalize (counts): bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, (hipStream_t)stream, partials, n_blocks, (long long)1, rows, C,
                     gamma, eps, momentum, running_mean, running_var, stats_out, scale, counts);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_bn_eval_scale(int C, const float* gamma, const float* running_var, float eps, float* scale,
                                    catseg_stream_t stream) {
  CS_REQUIRE(C > 0, "bn eval: C");
  hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, running_var, eps, scale);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_bn_apply_amax(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual,
                                    int ldr, float* z, int ldz, long long rows, int C, int relu, void* amax_record, catseg_stream_t stream);
extern "C" int catseg_bn_apply(const float* y, int ldy, const float* mean, const float* scale, const float* beta,
                               const float* residual, int ldr, float* z, int ldz, long long rows, int C, int relu,
                               catseg_stream_t stream) {
  return catseg_bn_apply_amax(y, ldy, mean, scale, beta, residual, ldr, z, ldz, rows, C, relu, nullptr, stream);
}
// the same, and max|z| folded into amax_record[0] (the 8-byte per-tensor record of catseg_dconv3_f16x2; zeroed by the caller; may be null)
extern "C" int catseg_bn_apply_amax(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual,
                                    int ldr, float* z, int ldz, long long rows, int C, int relu, void* amax_record, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0 && ldz % 4 == 0 && (residual == nullptr || ldr % 4 == 0),
             "bn apply: C and ld must be multiples of 4");
  CS_REQUIRE(cs_aligned16(y) && cs_aligned16(z) && cs_aligned16(mean) && cs_aligned16(scale) && cs_aligned16(beta) &&
                 cs_aligned16(residual), "bn apply: alignment");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, y, ldy, mean, scale,
                     beta, residual, ldr, z, ldz, rows, C, relu, (unsigned*)amax_record);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_bn_backward_amax(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* stats,
                                       const float* gamma, const float* beta, long long rows, int C, int relu, float* dy, int lddy,
                                       float* dgamma, float* dbeta, float* dres, int lddres, int dres_accumulate, void* workspace,
                                       size_t workspace_bytes, void* amax_record, catseg_stream_t stream);
extern "C" int catseg_bn_backward(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                                  const float* stats, const float* gamma, const float* beta, long long rows, int C, int relu,
                                  float* dy, int lddy, float* dgamma, float* dbeta, float* dres, int lddres,
                                  int dres_accumulate, void* workspace, size_t workspace_bytes,
                                  catseg_stream_t stream) {
  return catseg_bn_backward_amax(dz, lddz, z, ldz, y, ldy, stats, gamma, beta, rows, C, relu, dy, lddy, dgamma, dbeta, dres, lddres, dres_accumulate,
                                 workspace, workspace_bytes, nullptr, stream);
}
// the same, and max|dy| folded into amax_record[0] (may be null)
extern "C" int catseg_bn_backward_amax(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* stats,
                                       const float* gamma, const float* beta, long long rows, int C, int relu, float* dy, int lddy,
                                       float* dgamma, float* dbeta, float* dres, int lddres, int dres_accumulate, void* workspace,
                                       size_t workspace_bytes, void* amax_record, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && lddz % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0, "bn bwd: C and ld must be multiples of 4");
  CS_REQUIRE(!relu || (z != nullptr && ldz % 4 == 0) || (z == nullptr && beta != nullptr && dres == nullptr),
             "bn bwd: relu needs z, or (no residual branch) beta to recompute the mask from y");
  CS_REQUIRE(cs_aligned16(dz) && cs_aligned16(y) && cs_aligned16(dy) && cs_aligned16(stats) && cs_aligned16(gamma) &&
                 cs_aligned16(z) && cs_aligned16(dres), "bn bwd: alignment");
  if (workspace_bytes < catseg_bn_workspace(rows, C) || !workspace) {
    catseg_set_error("bn bwd: workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const RowSplit s = plan_rows(rows, C);
  float* part = (float*)workspace;
  float* coef = part + (size_t)kMaxRowBlocks * 3 * ((C + 3) & ~3);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(s.nrb, s.gy), dim3(256), 0, st, dz, lddz, z, ldz, y, ldy, stats, gamma, beta, rows, C, relu, s, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, (const float*)part, s.nrb, rows, C, dgamma, dbeta, coef);
#ifdef BN_AB_DOUBLE_FINALIZE   // (TIMING-ONLY build: the finalize launches issued twice; the added time = their cost in the step.  NOT result
                               //  preserving on the forward side: running_mean / running_var receive the momentum update twice)
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, (const float*)part, s.nrb, rows, C, dgamma, dbeta, coef);
#endif
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, st, dz, lddz, z, ldz, y, ldy, stats, gamma,
                     beta, (const float*)coef, rows, C, relu, dy, lddy, dres, lddres, dres_accumulate, (unsigned*)amax_record);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// backward of z = relu(bn(q)) when the producer of dz has already masked it (g = dz where z > 0) and left the per-block sums
// [n_blocks][2][C] of g and g * xhat (catseg_dconv3_bnbwd): the merge of the sums and the apply pass, i.e. catseg_bn_backward
// without its first pass over (dz, q)
extern "C" int catseg_bn_backward_pre_amax(const float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma,
                                           const float* partials, int n_blocks, long long rows, int C, float* dq, int lddq, float* dgamma,
                                           float* dbeta, void* workspace, size_t workspace_bytes, void* amax_record, catseg_stream_t stream);
extern "C" int catseg_bn_backward_pre(const float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma,
                                      const float* partials, int n_blocks, long long rows, int C, float* dq, int lddq, float* dgamma,
                                      float* dbeta, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  return catseg_bn_backward_pre_amax(g, ldg, q, ldq, stats, gamma, partials, n_blocks, rows, C, dq, lddq, dgamma, dbeta, workspace, workspace_bytes,
                                     nullptr, stream);
}
extern "C" int catseg_bn_backward_pre_amax(const float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma,
                                           const float* partials, int n_blocks, long long rows, int C, float* dq, int lddq, float* dgamma,
                                           float* dbeta, void* workspace, size_t workspace_bytes, void* amax_record, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && ldg % 4 == 0 && ldq % 4 == 0 && lddq % 4 == 0 && n_blocks > 0 && partials,
             "bn bwd (pre): C and ld must be multiples of 4");
  CS_REQUIRE(cs_aligned16(g) && cs_aligned16(q) && cs_aligned16(dq) && cs_aligned16(stats) && cs_aligned16(gamma), "bn bwd (pre): alignment");
  const size_t need = cs_align_up((size_t)2 * ((C + 3) & ~3) * 4, 256);
  if (workspace_bytes < need || !workspace) {
    catseg_set_error("bn bwd (pre): workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* coef = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, partials, n_blocks, rows, C, dgamma, dbeta, coef);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, st, g, ldg, (const float*)nullptr, 0, q, ldq, stats,
                     gamma, (const float*)nullptr, (const float*)coef, rows, C, 0, dq, lddq, (float*)nullptr, 0, 0, (unsigned*)amax_record);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}


// ---- producers of fp16 x 2 operand planes (csrc/planes.h; round 4) ---------------------------------------------------------------------------
// catseg_bn_finalize_counts + the bound of the normalised output: z_record[CS_REC_BOUND] = max over the channels of
// |scale| (max|y| + |mean|) + |beta|, with max|y| from y_record (the amax slots catseg_dconv3_pl's epilogue filled)
extern "C" int catseg_bn_finalize_counts_bound(const float* partials, int n_blocks, const int* counts, long long rows, int C, const float* gamma,
                                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                               float* stats_out, float* scale, const void* y_record, void* z_record, catseg_stream_t stream) {
  CS_REQUIRE(partials && counts && n_blocks > 0 && rows > 0 && C > 0 && beta && y_record && z_record, "bn finalize (bound): bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, (hipStream_t)stream, partials, n_blocks, (long long)1, rows, C,
                     gamma, eps, momentum, running_mean, running_var, stats_out, scale, counts, (const unsigned*)y_record, (unsigned*)z_record, beta);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_bn_apply that writes the planes of z (z_planes, catseg_planes_bytes(rows, C)) with the exponent derived from z_record's bound
// (+ max|residual| from residual_record), z itself only when z != NULL, and folds max|z| into z_record's amax slots
extern "C" int catseg_bn_apply_planes(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual,
                                      int ldr, const void* residual_record, float* z, int ldz, void* z_planes, long long rows, int C, int relu,
                                      void* z_record, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && ldy % 4 == 0 && (z == nullptr || ldz % 4 == 0) && (residual == nullptr || (ldr % 4 == 0 && residual_record)),
             "bn apply (planes): C must be a multiple of 8, ld of 4; a residual needs its amax record");
  CS_REQUIRE(cs_aligned16(y) && cs_aligned16(z) && cs_aligned16(mean) && cs_aligned16(scale) && cs_aligned16(beta) && cs_aligned16(residual) &&
                 cs_aligned16(z_planes) && z_planes && z_record, "bn apply (planes): alignment");
  const long long tiles = ((rows + 127) / 128) * ((C / 8 + 7) / 8);
  hipLaunchKernelGGL(bn_apply_planes_kernel, dim3((int)(tiles > 8192 ? 8192 : tiles)), dim3(256), 0, (hipStream_t)stream, y, ldy, mean, scale, beta,
                     residual, ldr, (const unsigned*)residual_record, z, ldz, (unsigned char*)z_planes, rows, C, relu, (unsigned*)z_record);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_bn_backward with dy written as planes ONLY.  g_record: a zeroed amax record, receives max|masked gradient| (first pass);
// y_record: max|y| of the forward pass; dy_record: zeroed, receives the bound, the exponent and nothing else
extern "C" int catseg_bn_backward_planes(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* stats,
                                         const float* gamma, const float* beta, long long rows, int C, int relu, void* dy_planes, void* dy_record,
                                         void* g_record, const void* y_record, float* dgamma, float* dbeta, float* dres, int lddres,
                                         int dres_accumulate, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && lddz % 4 == 0 && ldy % 4 == 0 && dy_planes && dy_record && g_record && y_record,
             "bn bwd (planes): C must be a multiple of 8, ld of 4; records required");
  CS_REQUIRE(!relu || (z != nullptr && ldz % 4 == 0) || (z == nullptr && beta != nullptr && dres == nullptr),
             "bn bwd (planes): relu needs z, or (no residual branch) beta to recompute the mask from y");
  CS_REQUIRE(cs_aligned16(dz) && cs_aligned16(y) && cs_aligned16(dy_planes) && cs_aligned16(stats) && cs_aligned16(gamma) && cs_aligned16(z) &&
                 cs_aligned16(dres), "bn bwd (planes): alignment");
  if (workspace_bytes < catseg_bn_workspace(rows, C) || !workspace) {
    catseg_set_error("bn bwd (planes): workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const RowSplit s = plan_rows(rows, C);
  float* part = (float*)workspace;
  float* coef = part + (size_t)kMaxRowBlocks * 3 * ((C + 3) & ~3);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(s.nrb, s.gy), dim3(256), 0, st, dz, lddz, z, ldz, y, ldy, stats, gamma, beta, rows, C, relu, s, part,
                     (unsigned*)g_record);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, (const float*)part, s.nrb, rows, C, dgamma, dbeta, coef,
                     (const unsigned*)g_record, (const unsigned*)y_record, stats, gamma, (unsigned*)dy_record);
  const long long tiles = ((rows + 127) / 128) * ((C / 8 + 7) / 8);
  hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3((int)(tiles > 8192 ? 8192 : tiles)), dim3(256), 0, st, dz, lddz, z, ldz, y, ldy, stats, gamma,
                     beta, (const float*)coef, rows, C, relu, (unsigned char*)dy_planes, (unsigned*)dy_record, dres, lddres, dres_accumulate);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_bn_backward_pre with dq written as planes only.  g_record: the amax record catseg_dconv3_pl_bnbwd filled with max|g|
extern "C" int catseg_bn_backward_pre_planes(const float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma,
                                             const float* partials, int n_blocks, long long rows, int C, void* dq_planes, void* dq_record,
                                             const void* g_record, const void* y_record, float* dgamma, float* dbeta, void* workspace,
                                             size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && ldg % 4 == 0 && ldq % 4 == 0 && n_blocks > 0 && partials && dq_planes && dq_record && g_record && y_record,
             "bn bwd (pre, planes): bad args");
  CS_REQUIRE(cs_aligned16(g) && cs_aligned16(q) && cs_aligned16(dq_planes) && cs_aligned16(stats) && cs_aligned16(gamma), "bn bwd (pre, planes): alignment");
  const size_t need = cs_align_up((size_t)2 * ((C + 3) & ~3) * 4, 256);
  if (workspace_bytes < need || !workspace) {
    catseg_set_error("bn bwd (pre, planes): workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* coef = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, partials, n_blocks, rows, C, dgamma, dbeta, coef,
                     (const unsigned*)g_record, (const unsigned*)y_record, stats, gamma, (unsigned*)dq_record);
  const long long tiles = ((rows + 127) / 128) * ((C / 8 + 7) / 8);
  hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3((int)(tiles > 8192 ? 8192 : tiles)), dim3(256), 0, st, g, ldg, (const float*)nullptr, 0, q, ldq,
                     stats, gamma, (const float*)nullptr, (const float*)coef, rows, C, 0, (unsigned char*)dq_planes, (unsigned*)dq_record,
                     (float*)nullptr, 0, 0);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_bn_backward with dy written ONLY as the blocked fp16 x 2 planes of catseg_split2h (dy_planes: catseg_split2h_blocked_elems(rows, C)
// halves; dy_scale: its 8-byte record), for a layer whose backward-weight / backward-data run catseg_conv2d_bwd_weight_f16x2_blocked /
// catseg_conv2d_bwd_data_f16x2_blocked; dbias (may be null) = the column sums of dy (the gradient of a bias in front of the BatchNorm).
// No residual branch.  g_record / y_record / dy_record: three zeroed amax records (csrc/common.h) -- they receive max|g|, max|y|, the bound of dy.
extern "C" size_t catseg_bn_backward_h2_workspace(long long rows, int C) {
  return catseg_bn_workspace(rows, C) + cs_align_up((size_t)256 * ((C + 3) & ~3) * 4, 256);
}
extern "C" int catseg_bn_backward_h2(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* stats,
                                     const float* gamma, const float* beta, long long rows, int C, int relu, void* dy_planes, void* dy_scale,
                                     float* dgamma, float* dbeta, float* dbias, void* g_record, void* y_record, void* dy_record, void* workspace,
                                     size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 64 == 0 && lddz % 4 == 0 && ldy % 4 == 0 && dy_planes && dy_scale && g_record && y_record && dy_record,
             "bn bwd (h2 planes): C must be a multiple of 64, ld of 4; planes, scale and records required");
  CS_REQUIRE(!relu || (z != nullptr && ldz % 4 == 0) || (z == nullptr && beta != nullptr),
             "bn bwd (h2 planes): relu needs z, or beta to recompute the mask from y");
  CS_REQUIRE(cs_aligned16(dz) && cs_aligned16(y) && cs_aligned16(dy_planes) && cs_aligned16(stats) && cs_aligned16(gamma) && cs_aligned16(z) &&
                 cs_aligned16(beta) && (((uintptr_t)dy_scale) & 7) == 0, "bn bwd (h2 planes): alignment");
  CS_REQUIRE(rows * C * 4 < (1ll << 32) - 64, "bn bwd (h2 planes): the two planes must stay below 4 GB");
  if (workspace_bytes < catseg_bn_backward_h2_workspace(rows, C) || !workspace) {
    catseg_set_error("bn bwd (h2 planes): workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const RowSplit s = plan_rows(rows, C);
  float* part = (float*)workspace;
  float* coef = part + (size_t)kMaxRowBlocks * 3 * ((C + 3) & ~3);
  float* colpart = (float*)((char*)workspace + catseg_bn_workspace(rows, C));
  unsigned* g_rec = (unsigned*)g_record;
  unsigned* y_rec = (unsigned*)y_record;
  unsigned* dy_rec = (unsigned*)dy_record;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(s.nrb, s.gy), dim3(256), 0, st, dz, lddz, z, ldz, y, ldy, stats, gamma, beta, rows, C, relu, s, part,
                     g_rec, y_rec);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, (const float*)part, s.nrb, rows, C, dgamma, dbeta, coef,
                     (const unsigned*)g_rec, (const unsigned*)y_rec, stats, gamma, dy_rec);
  const long long rblocks = (rows + 127) / 128;
  const int gx = (int)(rblocks < 256 ? rblocks : 256);
  hipLaunchKernelGGL(bn_bwd_apply_h2_kernel, dim3(gx, C / 64), dim3(256), 0, st, dz, lddz, z, ldz, y, ldy, stats, gamma, beta, (const float*)coef, rows,
                     C, relu, (unsigned char*)dy_planes, (long long)rows * C * 2, (const unsigned*)dy_rec, (unsigned*)dy_scale,
                     dbias ? colpart : (float*)nullptr);
  if (dbias) hipLaunchKernelGGL(colsum_rows_kernel, dim3((C + 63) / 64), dim3(256), 0, st, (const float*)colpart, gx, C, dbias);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// ---- the ReLU mask as bits (round 5): a residual block's output z = relu(bn(y) + residual) is read by its BatchNorm backward only for its
// sign.  Mask = rows x C / 8 bytes: bit (e & 7) of byte e >> 3 for the flat element index e = r C + c (C % 8 == 0).  catseg_bn_apply_mask /
// catseg_bn_apply_planes_mask = catseg_bn_apply_amax / catseg_bn_apply_planes (relu on) that also write it; catseg_bn_backward_mask /
// catseg_bn_backward_planes_mask = catseg_bn_backward_amax / catseg_bn_backward_planes reading the bits instead of z in both passes (4 bytes per
// element less each).  The residual BatchNorms of the trunk (models/HRNetv2.py:36-106) and of torchvision's blocks.
extern "C" size_t catseg_bn_mask_bytes(long long rows, int C) { return (size_t)(rows * C / 8); }

extern "C" int catseg_bn_apply_mask(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual,
                                    int ldr, float* z, int ldz, long long rows, int C, void* amax_record, void* mask, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && ldy % 4 == 0 && ldz % 4 == 0 && (residual == nullptr || ldr % 4 == 0) && mask,
             "bn apply (mask): C must be a multiple of 8, ld of 4");
  CS_REQUIRE(cs_aligned16(y) && cs_aligned16(z) && cs_aligned16(mean) && cs_aligned16(scale) && cs_aligned16(beta) && cs_aligned16(residual),
             "bn apply (mask): alignment");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, y, ldy, mean, scale, beta, residual, ldr, z,
                     ldz, rows, C, 1, (unsigned*)amax_record, (unsigned char*)mask);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_bn_backward_mask(const float* dz, int lddz, const void* mask, const float* y, int ldy, const float* stats, const float* gamma,
                                       long long rows, int C, float* dy, int lddy, float* dgamma, float* dbeta, float* dres, int lddres,
                                       int dres_accumulate, void* workspace, size_t workspace_bytes, void* amax_record, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && lddz % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && mask, "bn bwd (mask): C must be a multiple of 8, ld of 4");
  CS_REQUIRE(cs_aligned16(dz) && cs_aligned16(y) && cs_aligned16(dy) && cs_aligned16(stats) && cs_aligned16(gamma) && cs_aligned16(dres),
             "bn bwd (mask): alignment");
  if (workspace_bytes < catseg_bn_workspace(rows, C) || !workspace) {
    catseg_set_error("bn bwd (mask): workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const RowSplit s = plan_rows(rows, C);
  float* part = (float*)workspace;
  float* coef = part + (size_t)kMaxRowBlocks * 3 * ((C + 3) & ~3);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(s.nrb, s.gy), dim3(256), 0, st, dz, lddz, (const float*)nullptr, 0, y, ldy, stats, gamma,
                     (const float*)nullptr, rows, C, 1, s, part, (unsigned*)nullptr, (unsigned*)nullptr, (const unsigned char*)mask);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, (const float*)part, s.nrb, rows, C, dgamma, dbeta, coef);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, st, dz, lddz, (const float*)nullptr, 0, y, ldy, stats, gamma,
                     (const float*)nullptr, (const float*)coef, rows, C, 1, dy, lddy, dres, lddres, dres_accumulate, (unsigned*)amax_record,
                     (const unsigned char*)mask);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_bn_apply_planes (relu on) that also writes the mask
extern "C" int catseg_bn_apply_planes_mask(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual,
                                           int ldr, const void* residual_record, float* z, int ldz, void* z_planes, long long rows, int C,
                                           void* z_record, void* mask, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && ldy % 4 == 0 && (z == nullptr || ldz % 4 == 0) && (residual == nullptr || (ldr % 4 == 0 && residual_record)) && mask,
             "bn apply (planes, mask): C must be a multiple of 8, ld of 4; a residual needs its amax record");
  CS_REQUIRE(cs_aligned16(y) && cs_aligned16(z) && cs_aligned16(mean) && cs_aligned16(scale) && cs_aligned16(beta) && cs_aligned16(residual) &&
                 cs_aligned16(z_planes) && z_planes && z_record, "bn apply (planes, mask): alignment");
  const long long tiles = ((rows + 127) / 128) * ((C / 8 + 7) / 8);
  hipLaunchKernelGGL(bn_apply_planes_kernel, dim3((int)(tiles > 8192 ? 8192 : tiles)), dim3(256), 0, (hipStream_t)stream, y, ldy, mean, scale, beta,
                     residual, ldr, (const unsigned*)residual_record, z, ldz, (unsigned char*)z_planes, rows, C, 1, (unsigned*)z_record,
                     (unsigned char*)mask);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_bn_backward_planes (relu on) reading the mask instead of z
extern "C" int catseg_bn_backward_planes_mask(const float* dz, int lddz, const void* mask, const float* y, int ldy, const float* stats,
                                              const float* gamma, long long rows, int C, void* dy_planes, void* dy_record, void* g_record,
                                              const void* y_record, float* dgamma, float* dbeta, float* dres, int lddres, int dres_accumulate,
                                              void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && lddz % 4 == 0 && ldy % 4 == 0 && dy_planes && dy_record && g_record && y_record && mask,
             "bn bwd (planes, mask): C must be a multiple of 8, ld of 4; records required");
  CS_REQUIRE(cs_aligned16(dz) && cs_aligned16(y) && cs_aligned16(dy_planes) && cs_aligned16(stats) && cs_aligned16(gamma) && cs_aligned16(dres),
             "bn bwd (planes, mask): alignment");
  if (workspace_bytes < catseg_bn_workspace(rows, C) || !workspace) {
    catseg_set_error("bn bwd (planes, mask): workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const RowSplit s = plan_rows(rows, C);
  float* part = (float*)workspace;
  float* coef = part + (size_t)kMaxRowBlocks * 3 * ((C + 3) & ~3);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(s.nrb, s.gy), dim3(256), 0, st, dz, lddz, (const float*)nullptr, 0, y, ldy, stats, gamma,
                     (const float*)nullptr, rows, C, 1, s, part, (unsigned*)g_record, (unsigned*)nullptr, (const unsigned char*)mask);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, (const float*)part, s.nrb, rows, C, dgamma, dbeta, coef,
                     (const unsigned*)g_record, (const unsigned*)y_record, stats, gamma, (unsigned*)dy_record);
  const long long tiles = ((rows + 127) / 128) * ((C / 8 + 7) / 8);
  hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3((int)(tiles > 8192 ? 8192 : tiles)), dim3(256), 0, st, dz, lddz, (const float*)nullptr, 0, y, ldy,
                     stats, gamma, (const float*)nullptr, (const float*)coef, rows, C, 1, (unsigned char*)dy_planes, (unsigned*)dy_record, dres, lddres,
                     dres_accumulate, (const unsigned char*)mask);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// ---- the K-class classifier fused with the BatchNorm + ReLU in front of it (csrc/headfuse.h).
// Forward: logits[rows][ldl] = relu((y - mean) * scale + beta) Wh^T + bh, columns [K, zero_to) zeroed; Wh [K][C] is the 1 x 1 convolution's
// weight (models/OCR.py:74,97), mean / scale what catseg_bn_finalize left.  z is never written.
extern "C" int catseg_head_fwd(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* wh, const float* bh,
                               int K, long long rows, int C, float* logits, int ldl, int zero_to, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C >= 32 && C % 32 == 0 && C <= 512 && K >= 1 && K <= 32 && ldy % 4 == 0 && ldy >= C && ldl >= K && zero_to <= ldl &&
                 zero_to <= 32, "head forward: C a multiple of 32 up to 512, K <= 32, ld of y a multiple of 4");
  CS_REQUIRE(y && mean && scale && beta && wh && logits && cs_aligned16(y), "head forward: pointers / alignment");
  const size_t lds = (size_t)(C * 33 + 3 * C) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)hf_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (512 * 33 + 3 * 512) * 4) != hipSuccess) {
      catseg_set_error("head forward: cannot raise the dynamic LDS limit");
      return CATSEG_EHIP;
    }
    attr_set = true;
  }
  const long long groups = (rows + 31) / 32;
  const long long want = (groups + kHfFwdWaves - 1) / kHfFwdWaves;
  hipLaunchKernelGGL(hf_fwd_kernel, dim3((int)(want < 512 ? want : 512)), dim3(kHfFwdWaves * 64), lds, (hipStream_t)stream, y, ldy, mean, scale, beta, wh, bh, K,
                     rows, C, logits, ldl, zero_to);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// Backward of the same: from the gradient of the logits dl [rows][lddl] (lddl >= 32) to
//   dy_planes / dy_scale   the gradient of y as the blocked fp16 x 2 planes of catseg_bn_backward_h2 (same records: g / y / dy_record zeroed)
//   dgamma, dbeta          of the BatchNorm;  dbias (may be null): column sums of dy
//   dwh [K][C], dbh [K]    of the classifier (written, not accumulated; dbh may be null)
// row blocks of the backward launches: x (C / 128) channel blocks of 4 waves = 512 blocks, two per CU, all resident
static int head_blocks(int C) { return 512 / ((C + 127) / 128); }
extern "C" size_t catseg_head_backward_workspace(long long rows, int C) {
  const size_t hb = (size_t)head_blocks(C);
  return catseg_bn_workspace(rows, C) + cs_align_up(hb * C * 4, 256) + cs_align_up(hb * 32 * C * 4, 256) + cs_align_up(hb * 32 * 4, 256);
}
extern "C" int catseg_head_backward(const float* dl, int lddl, const float* y, int ldy, const float* stats, const float* gamma, const float* beta,
                                    const float* wh, int K, long long rows, int C, void* dy_planes, void* dy_scale, float* dgamma, float* dbeta,
                                    float* dbias, float* dwh, float* dbh, void* g_record, void* y_record, void* dy_record, void* workspace,
                                    size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C >= 64 && C % 64 == 0 && C <= 512 && K >= 1 && K <= 32 && lddl >= 32 && lddl % 4 == 0 && ldy >= C,
             "head backward: C a multiple of 64 up to 512, K <= 32, ld of the logits gradient >= 32 and a multiple of 4");
  CS_REQUIRE(dl && y && stats && gamma && beta && wh && dy_planes && dy_scale && dwh && g_record && y_record && dy_record && cs_aligned16(dl) &&
                 cs_aligned16(dy_planes) && (((uintptr_t)dy_scale) & 7) == 0, "head backward: pointers / alignment");
  CS_REQUIRE(rows * C * 4 < (1ll << 32) - 64 && (rows + 32 * 512) * (long long)ldy * 4 < (1ll << 32) && (rows + 32 * 512) * (long long)lddl * 4 < (1ll << 32),
             "head backward: planes and operands must stay below 4 GB (32-bit buffer offsets)");
  if (workspace_bytes < catseg_head_backward_workspace(rows, C) || !workspace) {
    catseg_set_error("head backward: workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* coef = part + (size_t)kMaxRowBlocks * 3 * ((C + 3) & ~3);
  char* w = (char*)workspace + catseg_bn_workspace(rows, C);
  float* colpart = (float*)w;
  const size_t hb = (size_t)head_blocks(C);
  w += cs_align_up(hb * C * 4, 256);
  float* dws = (float*)w;
  w += cs_align_up(hb * 32 * C * 4, 256);
  float* dbs = (float*)w;
  const long long chunks = (rows + 31) / 32;
  const int nb = (int)(chunks < (long long)hb ? chunks : (long long)hb);
  HfBwdArgs a;
  a.dl = dl; a.lddl = lddl; a.y = y; a.ldy = ldy; a.stats = stats; a.gamma = gamma; a.beta = beta; a.wh = wh; a.K = K; a.rows = rows; a.C = C;
  a.part = part; a.dws = dws; a.dbs = dbs; a.g_rec = (unsigned*)g_record; a.y_rec = (unsigned*)y_record;
  a.coef = coef; a.planes = (unsigned char*)dy_planes; a.plane_bytes = rows * C * 2; a.dy_rec = (const unsigned*)dy_record;
  a.scale = (unsigned*)dy_scale; a.colpart = dbias ? colpart : nullptr;
  hipLaunchKernelGGL((hf_bwd_kernel<false, 1, 4>), dim3(nb, (C + 127) / 128), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(1024), 0, st, (const float*)part, nb, rows, C, dgamma, dbeta, coef,
                     (const unsigned*)a.g_rec, (const unsigned*)a.y_rec, stats, gamma, (unsigned*)dy_record);
  hipLaunchKernelGGL(hf_reduce_kernel, dim3((K * C + K + 63) / 64), dim3(256), 0, st, (const float*)dws, (const float*)dbs, nb, K, C, dwh, dbh);
  hipLaunchKernelGGL((hf_bwd_kernel<true, 1, 4>), dim3(nb, (C + 127) / 128), dim3(256), 0, st, a);
  if (dbias) hipLaunchKernelGGL(colsum_rows_kernel, dim3((C + 63) / 64), dim3(256), 0, st, (const float*)colpart, nb, C, dbias);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
