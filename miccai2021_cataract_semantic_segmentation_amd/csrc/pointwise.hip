// HBM-bound companions of the convolutions: layout transforms, max-pool, bilinear resize,
// global average pool and the two softmaxes of the OCR head.  All NHWC fp32, all
// deterministic (backward passes are written as gathers, never atomics).
#include "planes.h"

// 1 (default): catseg_bilinear_bwd runs as one launch with its intermediate row in LDS where the layout allows; 0: the two separable passes
int catseg_g_bilinear_bwd_fused = 1;
extern "C" int catseg_debug_set_bilinear_bwd_fused(int on) {
  catseg_g_bilinear_bwd_fused = on ? 1 : 0;
  return CATSEG_OK;
}

namespace {

int grid_for(long long total, int per_block = 256) {
  long long b = (total + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// ------------------------------------------------------------------ layout
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int B, long long HW) {
  const long long total = (long long)B * HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / HW, p = i - b * HW;
    const float* s = x + b * 3 * HW + p;
    f32x4 v = {s[0], s[HW], s[2 * HW], 0.f};
    *(f32x4*)(y + i * 4) = v;
  }
}

// w [O][7][7][3] (OHWI) -> packed [O][7][8][4], zero padded
__global__ void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ pk, int O) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O * 7 * 8 * 4) return;
  const int c = i & 3, kx = (i >> 2) & 7, ky = (i >> 5) % 7, o = i / (7 * 32);
  pk[i] = (c < 3 && kx < 7) ? w[((o * 7 + ky) * 7 + kx) * 3 + c] : 0.f;
}
__global__ void stem_unpack_kernel(const float* __restrict__ pk, float* __restrict__ dw, int O) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O * 147) return;
  const int c = i % 3, kx = (i / 3) % 7, ky = (i / 21) % 7, o = i / 147;
  dw[i] = pk[((o * 7 + ky) * 8 + kx) * 4 + c];
}

__global__ void axpy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, long long rows, int C,
                              float alpha, int acc) {
  const int cpt = C >> 2;
  CS_QUAD_LOOP(rows, cpt, r, c) {
    f32x4 v = *(const f32x4*)(src + r * lds + c) * alpha;
    f32x4* d = (f32x4*)(dst + r * ldd + c);
    *d = acc ? (*d + v) : v;
  }
}

// x *= s[0]  (s lives on the device: the upstream gradient of a loss, no host sync)
__global__ void scale_dev_kernel(float* __restrict__ x, long long n, const float* __restrict__ s) {
  const float a = s[0];
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4* p = (f32x4*)x + i;
    *p = *p * a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[(n4 << 2) + threadIdx.x] *= a;
}

// out = act(sum_i in_i)  (HRNet fuse: up to 4 same-resolution terms), all [rows][ld_i]
struct AddArgs {
  const float* in[4];
  int ld[4];
  int n;
};
__global__ void add_n_act_kernel(AddArgs a, float* __restrict__ out, int ldo, long long rows, int C, int relu, unsigned* __restrict__ amax) {
  const int cpt = C >> 2;
  unsigned m = 0;
  CS_QUAD_LOOP(rows, cpt, r, c) {
    f32x4 v = *(const f32x4*)(a.in[0] + r * a.ld[0] + c);
    for (int k = 1; k < a.n; ++k) v += *(const f32x4*)(a.in[k] + r * a.ld[k] + c);
    if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    *(f32x4*)(out + r * ldo + c) = v;
    m = max(m, cs_abs_bits4(v));
  }
  if (amax) cs_amax_commit(m, amax);
}
// the same sum writing out AND its fp16 x 2 planes (csrc/planes.h): |sum| <= sum of the terms' max| | (their amax records), known before the pass
struct AddRecs { const unsigned* rec[4]; };
__global__ __launch_bounds__(256) void add_n_act_planes_kernel(AddArgs a, AddRecs rr, float* __restrict__ out, int ldo, unsigned char* __restrict__ planes,
                                                               long long rows, int C, int relu, unsigned* __restrict__ rec) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[CsPlaneTile::BYTES];
  float bound = 0.f;
  for (int k = 0; k < a.n; ++k) bound += __uint_as_float(cs_amax_read(rr.rec[k]));
  const int e = cs_plane_exponent(__float_as_uint(bound * 1.001f));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ((int*)rec)[CS_REC_EXP] = e;
    rec[CS_REC_FINAL] = __float_as_uint(bound * 1.001f);
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
        f32x4 v0 = *(const f32x4*)(a.in[0] + r * a.ld[0] + c), v1 = *(const f32x4*)(a.in[0] + r * a.ld[0] + c + 4);
        for (int k = 1; k < a.n; ++k) {
          v0 += *(const f32x4*)(a.in[k] + r * a.ld[k] + c);
          v1 += *(const f32x4*)(a.in[k] + r * a.ld[k] + c + 4);
        }
        if (relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { v0[k] = fmaxf(v0[k], 0.f); v1[k] = fmaxf(v1[k], 0.f); }
        }
        *(f32x4*)(out + r * ldo + c) = v0;
        *(f32x4*)(out + r * ldo + c + 4) = v1;
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
  cs_amax_commit(m, rec);
}
// g = dz * (z > 0)
__global__ void relu_bwd_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ z, int ldz, float* __restrict__ g, int ldg,
                                long long rows, int C) {
  const int cpt = C >> 2;
  CS_QUAD_LOOP(rows, cpt, r, c) {
    f32x4 d = *(const f32x4*)(dz + r * lddz + c);
    const f32x4 zz = *(const f32x4*)(z + r * ldz + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = zz[k] > 0.f ? d[k] : 0.f;
    *(f32x4*)(g + r * ldg + c) = d;
  }
}
// conv weight [O][T][cin] <-> zero-padded [O][T][cpad]
__global__ void weight_pad_kernel(const float* __restrict__ w, float* __restrict__ pk, int n, int cin, int cpad, int unpad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!unpad) {
    const int c = i % cpad, ot = i / cpad;
    pk[i] = c < cin ? w[ot * cin + c] : 0.f;
  } else {
    const int c = i % cin, ot = i / cin;
    pk[i] = w[ot * cpad + c];
  }
}

// ------------------------------------------------------------------ maxpool 3x3 / 2 / pad 1
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, uint8_t* __restrict__ idx,
                                   int B, int H, int W, int C, int Ho, int Wo) {
  const int cpt = C >> 2;
  const long long total = (long long)B * Ho * Wo * cpt;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / cpt;
    const int c = (int)(i - p * cpt) * 4;
    const int ox = (int)(p % Wo);
    const int oy = (int)((p / Wo) % Ho);
    const int b = (int)(p / ((long long)Wo * Ho));
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first[4] = {true, true, true, true};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = *(const f32x4*)(x + (((long long)b * H + iy) * W + ix) * ldx + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // ATen keeps the FIRST maximum in scan order (val > max, or NaN)
          if (first[k] || v[k] > best[k] || v[k] != v[k]) {
            best[k] = v[k];
            bi[k] = ky * 3 + kx;
            first[k] = false;
          }
        }
      }
    }
    *(f32x4*)(y + p * ldy + c) = best;
    *(uint32_t*)(idx + p * C + c) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
  }
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                   int lddx, int B, int H, int W, int C, int Ho, int Wo) {
  const int cpt = C >> 2;
  const long long total = (long long)B * H * W * cpt;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / cpt;
    const int c = (int)(i - p * cpt) * 4;
    const int ix = (int)(p % W);
    const int iy = (int)((p / W) % H);
    const int b = (int)(p / ((long long)W * H));
    f32x4 g = {0, 0, 0, 0};
    // output windows covering (iy, ix): oy in {(iy+1)/2 (ky = iy+1-2oy)}, with ky in 0..2
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy + 1 - ky;
      if (ty < 0 || (ty & 1)) continue;
      const int oy = ty >> 1;
      if (oy >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix + 1 - kx;
        if (tx < 0 || (tx & 1)) continue;
        const int ox = tx >> 1;
        if (ox >= Wo) continue;
        const long long q = ((long long)b * Ho + oy) * Wo + ox;
        const uint32_t pk = *(const uint32_t*)(idx + q * C + c);
        const f32x4 d = *(const f32x4*)(dy + q * lddy + c);
        const uint32_t me = ky * 3 + kx;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (((pk >> (8 * k)) & 0xff) == me) g[k] += d[k];
      }
    }
    *(f32x4*)(dx + p * lddx + c) = g;
  }
}

// ------------------------------------------------------------------ bilinear
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
// ATen's area_pixel_compute_source_index in fp32
__device__ __forceinline__ float src_index(float scale, int dst, bool align) {
  if (align) return scale * dst;
  const float s = scale * (dst + 0.5f) - 0.5f;
  return s < 0.f ? 0.f : s;
}
__device__ __forceinline__ void lerp_setup(float scale, int dst, bool align, int in_size, int& i0, int& i1, float& l0, float& l1) {
  const float s = src_index(scale, dst, align);
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - i0;
  l0 = 1.f - l1;
}
__host__ __device__ inline float resize_scale(int in, int out, bool align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}

// MODE 0: scalar (any C / ld); MODE 1: contiguous output rows (ldy == C), 4 consecutive floats of the flat (ox, c) index per
// thread, 16-byte stores (the K-class logits at full resolution: C = 25); MODE 2: C % 4 == 0, 4 channels of one pixel per thread,
// 16-byte loads and stores (HRNet / ASPP / decoder feature maps, also into channel slices of a concat buffer).
// The per-element arithmetic is the same expression in all modes (bit-identical results).
template <int MODE>
__global__ void bilinear_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int B, int H, int W, int C,
                                    int Ho, int Wo, int align, float sh, float sw, int acc) {
  // one block per output row (b, oy)
  const int b = blockIdx.x / Ho, oy = blockIdx.x - b * Ho;
  int y0, y1;
  float ly0, ly1;
  lerp_setup(sh, oy, align, H, y0, y1, ly0, ly1);
  const float* r0 = x + ((long long)b * H + y0) * W * ldx;
  const float* r1 = x + ((long long)b * H + y1) * W * ldx;
  float* o = y + ((long long)b * Ho + oy) * Wo * ldy;
  if (MODE == 0) {
    const int n = Wo * C;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int ox = i / C, c = i - ox * C;
      int x0, x1;
      float lx0, lx1;
      lerp_setup(sw, ox, align, W, x0, x1, lx0, lx1);
      const float v = ly0 * (lx0 * r0[x0 * ldx + c] + lx1 * r0[x1 * ldx + c]) + ly1 * (lx0 * r1[x0 * ldx + c] + lx1 * r1[x1 * ldx + c]);
      float* d = o + (long long)ox * ldy + c;
      *d = acc ? (*d + v) : v;
    }
  } else if (MODE == 1) {
    const int n4 = (Wo * C) >> 2;
    for (int u = threadIdx.x; u < n4; u += blockDim.x) {
      int ox = (u * 4) / C, c = u * 4 - ox * C;
      int x0, x1;
      float lx0, lx1;
      lerp_setup(sw, ox, align, W, x0, x1, lx0, lx1);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = ly0 * (lx0 * r0[x0 * ldx + c] + lx1 * r0[x1 * ldx + c]) + ly1 * (lx0 * r1[x0 * ldx + c] + lx1 * r1[x1 * ldx + c]);
        if (++c == C) {
          c = 0;
          ++ox;
          if (e < 3) lerp_setup(sw, ox < Wo ? ox : Wo - 1, align, W, x0, x1, lx0, lx1);
        }
      }
      f32x4* d = (f32x4*)(o + u * 4);
      *d = acc ? (*d + v) : v;
    }
  } else {
    const int c4n = C >> 2, n = Wo * c4n;
    for (int u = threadIdx.x; u < n; u += blockDim.x) {
      const int ox = u / c4n, c = (u - ox * c4n) * 4;
      int x0, x1;
      float lx0, lx1;
      lerp_setup(sw, ox, align, W, x0, x1, lx0, lx1);
      const f32x4 a0 = ld4(r0 + x0 * ldx + c), a1 = ld4(r0 + x1 * ldx + c), b0 = ld4(r1 + x0 * ldx + c), b1 = ld4(r1 + x1 * ldx + c);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ly0 * (lx0 * a0[e] + lx1 * a1[e]) + ly1 * (lx0 * b0[e] + lx1 * b1[e]);
      f32x4* d = (f32x4*)(o + (long long)ox * ldy + c);
      *d = acc ? (*d + v) : v;
    }
  }
}

// MODE 1 with the two source rows of an output row staged in LDS (round 6): every output element gathers four source values, 16 scalar loads per
// 16-byte store -- at the K-class logits (C = 25, 8 x 136 x 240 -> 8 x 544 x 960: 418 MB written) the texture path, not HBM, set the pace
// (228 us = 1.8 TB/s).  The rows are W x ldx floats each, contiguous and 16-byte aligned; the arithmetic is bilinear_fwd_kernel's expression on
// the same operands: bit-identical output.  Requires 2 W ldx floats of LDS (<= 64 KB: two blocks per CU), ldx % 4 == 0, x 16-byte aligned.
__global__ __launch_bounds__(256) void bilinear_fwd_lds_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int B, int H, int W, int C,
                                                               int Ho, int Wo, int align, float sh, float sw, int acc) {
  extern __shared__ __attribute__((aligned(16))) float bl_rows[];
  const int b = blockIdx.x / Ho, oy = blockIdx.x - b * Ho;
  int y0, y1;
  float ly0, ly1;
  lerp_setup(sh, oy, align, H, y0, y1, ly0, ly1);
  const int rowf = W * ldx;
  const f32x4* g0 = (const f32x4*)(x + ((long long)b * H + y0) * rowf);
  const f32x4* g1 = (const f32x4*)(x + ((long long)b * H + y1) * rowf);
  f32x4* s0 = (f32x4*)bl_rows;
  f32x4* s1 = (f32x4*)(bl_rows + rowf);
  for (int i = threadIdx.x; i < (rowf >> 2); i += 256) {
    s0[i] = g0[i];
    s1[i] = g1[i];
  }
  __syncthreads();
  const float* r0 = bl_rows;
  const float* r1 = bl_rows + rowf;
  float* o = y + ((long long)b * Ho + oy) * Wo * C;
  const int n4 = (Wo * C) >> 2;
  for (int u = threadIdx.x; u < n4; u += 256) {
    int ox = (u * 4) / C, c = u * 4 - ox * C;
    int x0, x1;
    float lx0, lx1;
    lerp_setup(sw, ox, align, W, x0, x1, lx0, lx1);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = ly0 * (lx0 * r0[x0 * ldx + c] + lx1 * r0[x1 * ldx + c]) + ly1 * (lx0 * r1[x0 * ldx + c] + lx1 * r1[x1 * ldx + c]);
      if (++c == C) {
        c = 0;
        ++ox;
        if (e < 3) lerp_setup(sw, ox < Wo ? ox : Wo - 1, align, W, x0, x1, lx0, lx1);
      }
    }
    f32x4* d = (f32x4*)(o + u * 4);
    *d = acc ? (*d + v) : v;
  }
}

// backward pass 1: tmp[b, iy, ox, c] = sum_{oy} wy(oy -> iy) * dy[b, oy, ox, c]
// The weights of the candidate output rows depend only on (iy, oy): computed once per block.  MODE as above (1: lddy == C).
template <int MODE>
__global__ void bilinear_bwd_rows_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ tmp, int B, int H, int C, int Ho,
                                         int Wo, int align, float sh) {
  const int b = blockIdx.x / H, iy = blockIdx.x - b * H;
  // candidate output rows: source index in (iy-1, iy+1)
  int lo, hi;
  if (sh > 0.f) {
    lo = align ? (int)floorf((iy - 1) / sh) - 1 : (int)floorf((iy - 0.5f) / sh) - 2;
    hi = align ? (int)ceilf((iy + 1) / sh) + 1 : (int)ceilf((iy + 1.5f) / sh) + 1;
  } else {
    lo = 0; hi = Ho - 1;
  }
  if (lo < 0) lo = 0;
  if (hi > Ho - 1) hi = Ho - 1;
  auto weight = [&](int oy) {
    int y0, y1;
    float l0, l1;
    lerp_setup(sh, oy, align, H, y0, y1, l0, l1);
    float w = 0.f;
    if (y0 == iy) w += l0;
    if (y1 == iy) w += l1;
    return w;
  };
  const int n = Wo * C;
  float* trow = tmp + ((long long)blockIdx.x * Wo) * C;
  if (MODE == 0) {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
      const int ox = i / C, c = i - ox * C;
      float s = 0.f;
      for (int oy = lo; oy <= hi; ++oy) {
        const float w = weight(oy);
        if (w != 0.f) s = __builtin_fmaf(w, dy[(((long long)b * Ho + oy) * Wo + ox) * lddy + c], s);
      }
      trow[i] = s;
    }
  } else {
    // (the weights are wave-uniform: the branch on w != 0 is not divergent)
    const int c4n = C >> 2;
    const int n4 = MODE == 1 ? n >> 2 : Wo * c4n;
    for (int u = blockIdx.y * blockDim.x + threadIdx.x; u < n4; u += gridDim.y * blockDim.x) {
      long long off;
      if (MODE == 1) {
        off = (long long)u * 4;
      } else {
        const int ox = u / c4n;
        off = (long long)ox * lddy + (u - ox * c4n) * 4;
      }
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int oy = lo; oy <= hi; ++oy) {
        const float w = weight(oy);
        if (w != 0.f) {
          const f32x4 v = ld4(dy + ((long long)b * Ho + oy) * Wo * lddy + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] = __builtin_fmaf(w, v[e], s[e]);
        }
      }
      *(f32x4*)(trow + (long long)u * 4) = s;
    }
  }
}
// backward pass 2: dx[b, iy, ix, c] = sum_{ox} wx(ox -> ix) * tmp[b, iy, ox, c]
__global__ void bilinear_bwd_cols_kernel(const float* __restrict__ tmp, float* __restrict__ dx, int lddx, int W, int C, int Wo, int align,
                                         float sw, int zero_to, int acc) {
  const long long row = blockIdx.x;  // (b, iy)
  const float* t = tmp + row * Wo * C;
  float* o = dx + row * W * lddx;
  const int cw = zero_to > C ? zero_to : C;
  const int n = W * cw;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
    const int ix = i / cw, c = i - ix * cw;
    float* d = o + (long long)ix * lddx + c;
    if (c >= C) {
      *d = 0.f;
      continue;
    }
    int lo, hi;
    if (sw > 0.f) {
      lo = align ? (int)floorf((ix - 1) / sw) - 1 : (int)floorf((ix - 0.5f) / sw) - 2;
      hi = align ? (int)ceilf((ix + 1) / sw) + 1 : (int)ceilf((ix + 1.5f) / sw) + 1;
    } else {
      lo = 0; hi = Wo - 1;
    }
    if (lo < 0) lo = 0;
    if (hi > Wo - 1) hi = Wo - 1;
    float s = 0.f;
    for (int ox = lo; ox <= hi; ++ox) {
      int x0, x1;
      float l0, l1;
      lerp_setup(sw, ox, align, W, x0, x1, l0, l1);
      float w = 0.f;
      if (x0 == ix) w += l0;
      if (x1 == ix) w += l1;
      if (w != 0.f) s = __builtin_fmaf(w, t[ox * C + c], s);
    }
    *d = acc ? (*d + s) : s;
  }
}

// backward in ONE launch (round 6): the two separable passes above with the intermediate row `tmp[b, iy, ox, :]` kept in LDS instead of HBM.
// Block = (image b, input row iy, a segment of input columns, a channel chunk): phase 1 computes t[ox][c] = sum_oy wy(oy -> iy) dy[b, oy, ox, c] for
// the output columns the segment's input columns can gather from (16-byte loads along the contiguous (ox, c) index, exactly
// bilinear_bwd_rows_kernel's expression and order), phase 2 dx[b, iy, ix, c] = sum_ox wx(ox -> ix) t[ox][c] (bilinear_bwd_cols_kernel's): the
// results are BIT-IDENTICAL to the two-pass route, the 4 B per (input row x output column x channel) of tmp are neither written nor re-read
// (the K-class logits at 544 x 960 from 136 x 240: 104 MB each way per call), and one launch replaces two.
// Input rows are dealt to the blocks so that the rows iy, iy + 1, ... of one image go to ONE XCD (blocks b, b + 8, ... share an XCD's L2):
// every dy row feeds two input rows, the second read is an L2 hit instead of a second trip over the fabric.
// MODE 1: lddy == C, (Wo C) % 4 == 0 (contiguous output rows: the logits); MODE 2: C % 4 == 0, channel chunks of CC (feature maps, channel
// slices of a concatenation buffer).
constexpr int BIL_MAXROWS = 64;   // output rows that can weigh into one input row (the host falls back to two passes beyond)
struct BilinearBwdFused {
  const float* dy; int lddy;
  float* dx; int lddx;
  int B, H, W, C, Ho, Wo, align;
  float sh, sw;
  int zero_to, acc;
  int seg, nseg, CC, nchunk;      // input columns per block, segments per row, channels per chunk, chunks
};

__device__ __forceinline__ void bil_cand(float scale, int i, int align, int out_size, int& lo, int& hi) {
  if (scale > 0.f) {
    lo = align ? (int)floorf((i - 1) / scale) - 1 : (int)floorf((i - 0.5f) / scale) - 2;
    hi = align ? (int)ceilf((i + 1) / scale) + 1 : (int)ceilf((i + 1.5f) / scale) + 1;
  } else {
    lo = 0; hi = out_size - 1;
  }
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

template <int MODE>
__global__ __launch_bounds__(256) void bilinear_bwd_fused_kernel(const BilinearBwdFused a) {
  extern __shared__ __attribute__((aligned(16))) float t[];
  // XCD-aware row order: logical row = the (bid / 8)-th row of XCD (bid % 8)'s contiguous share
  const int nrow = gridDim.x, bid = blockIdx.x;
  const int q8 = nrow >> 3, r8 = nrow & 7, xcd = bid & 7;
  const int row = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int b = row / a.H, iy = row - b * a.H;
  const int sg = blockIdx.y / a.nchunk, ch = blockIdx.y - sg * a.nchunk;
  const int ix0 = sg * a.seg, nix = min(a.seg, a.W - ix0);
  const int c0 = ch * a.CC, cc = min(a.CC, a.C - c0);
  int ylo, yhi, oxa, oxb, tmp_;
  bil_cand(a.sh, iy, a.align, a.Ho, ylo, yhi);
  bil_cand(a.sw, ix0, a.align, a.Wo, oxa, tmp_);
  bil_cand(a.sw, ix0 + nix - 1, a.align, a.Wo, tmp_, oxb);
  auto wy = [&](int oy) {
    int y0, y1;
    float l0, l1;
    lerp_setup(a.sh, oy, a.align, a.H, y0, y1, l0, l1);
    float w = 0.f;
    if (y0 == iy) w += l0;
    if (y1 == iy) w += l1;
    return w;
  };
  // the output rows with a non-zero weight for this input row, in ascending order (what the two-pass kernel's `if (w != 0.f)` visits): found
  // once per block, so that the sweep below has no branch between its loads -- four rows' 16-byte loads are in flight per thread instead of
  // one load -> FMA round trip per row (the first version of this kernel was latency-bound at 2.5 TB/s)
  __shared__ float wl[BIL_MAXROWS];
  __shared__ int yl[BIL_MAXROWS];
  __shared__ int nw_s;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float w = wy(oy);
      if (w != 0.f && n < BIL_MAXROWS) {
        wl[n] = w;
        yl[n] = oy;
        ++n;
      }
    }
    nw_s = n;
  }
  __syncthreads();
  const int nw = nw_s;
  const float* drow0 = a.dy + ((long long)b * a.Ho) * a.Wo * a.lddy;
  const long long rstride = (long long)a.Wo * a.lddy;
  auto sweep = [&](const long long off) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= nw; k += 4) {
      const f32x4 v0 = ld4(drow0 + yl[k] * rstride + off), v1 = ld4(drow0 + yl[k + 1] * rstride + off);
      const f32x4 v2 = ld4(drow0 + yl[k + 2] * rstride + off), v3 = ld4(drow0 + yl[k + 3] * rstride + off);
      const float w0 = wl[k], w1 = wl[k + 1], w2 = wl[k + 2], w3 = wl[k + 3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] = __builtin_fmaf(w0, v0[e], s[e]);      // (explicit FMAs here and in the two-pass kernels: the same roundings by construction)
        s[e] = __builtin_fmaf(w1, v1[e], s[e]);
        s[e] = __builtin_fmaf(w2, v2[e], s[e]);
        s[e] = __builtin_fmaf(w3, v3[e], s[e]);
      }
    }
    for (; k < nw; ++k) {
      const f32x4 v = ld4(drow0 + yl[k] * rstride + off);
      const float w = wl[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = __builtin_fmaf(w, v[e], s[e]);
    }
    return s;
  };
  int f0 = 0;                       // MODE 1: first float of the (ox, c) run held in LDS
  if (MODE == 1) {
    f0 = (oxa * a.C) & ~3;
    int f1 = ((oxb + 1) * a.C + 3) & ~3;
    if (f1 > a.Wo * a.C) f1 = a.Wo * a.C;
    const int n4 = (f1 - f0) >> 2;
    for (int u = threadIdx.x; u < n4; u += 256) *(f32x4*)(t + 4 * u) = sweep(f0 + 4 * u);
  } else {
    const int q4 = cc >> 2, n4 = (oxb - oxa + 1) * q4;
    for (int u = threadIdx.x; u < n4; u += 256) {
      const int oxi = u / q4, q = u - oxi * q4;
      *(f32x4*)(t + oxi * a.CC + 4 * q) = sweep((long long)(oxa + oxi) * a.lddy + c0 + 4 * q);
    }
  }
  __syncthreads();
  // phase 2: this block's input columns x (its channels + the zero padding behind the last chunk)
  const bool last = ch == a.nchunk - 1;
  const int cw = (last && a.zero_to > a.C) ? cc + (a.zero_to - a.C) : cc;
  float* orow = a.dx + ((long long)row * a.W + ix0) * a.lddx + c0;
  for (int i = threadIdx.x; i < nix * cw; i += 256) {
    const int ixl = i / cw, c = i - ixl * cw;
    float* d = orow + (long long)ixl * a.lddx + c;
    if (c >= cc) {
      *d = 0.f;
      continue;
    }
    const int ix = ix0 + ixl;
    int lo, hi;
    bil_cand(a.sw, ix, a.align, a.Wo, lo, hi);
    float s = 0.f;
    for (int ox = lo; ox <= hi; ++ox) {
      int x0, x1;
      float l0, l1;
      lerp_setup(a.sw, ox, a.align, a.W, x0, x1, l0, l1);
      float w = 0.f;
      if (x0 == ix) w += l0;
      if (x1 == ix) w += l1;
      if (w != 0.f) s = __builtin_fmaf(w, MODE == 1 ? t[ox * a.C + c - f0] : t[(ox - oxa) * a.CC + c], s);
    }
    *d = a.acc ? (*d + s) : s;
  }
}

// host side of the fused launch: the candidate ranges exactly as the kernel computes them (same float expressions)
static void bil_cand_host(float scale, int i, int align, int out_size, int& lo, int& hi) {
  if (scale > 0.f) {
    lo = align ? (int)floorf((i - 1) / scale) - 1 : (int)floorf((i - 0.5f) / scale) - 2;
    hi = align ? (int)ceilf((i + 1) / scale) + 1 : (int)ceilf((i + 1.5f) / scale) + 1;
  } else {
    lo = 0; hi = out_size - 1;
  }
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

// LDS floats a block of `seg` input columns needs at most (0: does not fit `cap`)
static int bil_fused_lds_floats(int W, int Wo, int C, int CC, int mode, int align, float sw, int seg, int cap) {
  int need = 0;
  for (int ix0 = 0; ix0 < W; ix0 += seg) {
    const int nix = seg < W - ix0 ? seg : W - ix0;
    int oxa, oxb, t_;
    bil_cand_host(sw, ix0, align, Wo, oxa, t_);
    bil_cand_host(sw, ix0 + nix - 1, align, Wo, t_, oxb);
    int n;
    if (mode == 1) {
      const int f0 = (oxa * C) & ~3;
      int f1 = ((oxb + 1) * C + 3) & ~3;
      if (f1 > Wo * C) f1 = Wo * C;
      n = f1 - f0;
    } else {
      n = (oxb - oxa + 1) * CC;
    }
    if (n > cap) return 0;
    need = n > need ? n : need;
  }
  return need;
}

// ------------------------------------------------------------------ global average pool
__global__ void gap_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int HW, int C) {
  // grid (C/64 ceil, B); block 256 = 64 channels x 4 row lanes
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, b = blockIdx.y;
  float s = 0.f;
  if (c < C)
    for (int p = rl; p < HW; p += 4) s += x[((long long)b * HW + p) * ldx + c];
  __shared__ float sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < C) y[(long long)b * C + c] = (sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192]) / (float)HW;
}
// The same sum with 16-byte loads, 64 row lanes per channel quad and four independent load chains per thread: the scalar kernel above walks
// HW / 4 rows per thread through ONE dependent accumulator (2040 serial loads at 8 x 68 x 120 x 2048: 0.85 ms for 535 MB, the ASPP image-pool
// branch of DeepLabv3+).  Block = 16 channel quads x 64 row lanes; the row lanes are combined in a fixed order (shuffles over the four
// lanes of a wave, then the 16 waves in turn): deterministic.  Requires C % 4 == 0, ldx % 4 == 0, x 16-byte aligned.
__global__ __launch_bounds__(1024) void gap_fwd4_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int HW, int C) {
  const int g = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 64 + g * 4, b = blockIdx.y;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (c < C) {
    const float* base = x + (long long)b * HW * ldx + c;
    int p = rl;
    for (; p + 192 < HW; p += 256) {
      a0 += *(const f32x4*)(base + (long long)p * ldx);
      a1 += *(const f32x4*)(base + (long long)(p + 64) * ldx);
      a2 += *(const f32x4*)(base + (long long)(p + 128) * ldx);
      a3 += *(const f32x4*)(base + (long long)(p + 192) * ldx);
    }
    for (; p < HW; p += 64) a0 += *(const f32x4*)(base + (long long)p * ldx);
  }
  f32x4 s = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s[k] += __shfl_xor(s[k], 16, 64);
    s[k] += __shfl_xor(s[k], 32, 64);
  }
  __shared__ f32x4 sh[16][16];
  if ((threadIdx.x & 63) < 16) sh[threadIdx.x >> 6][g] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    f32x4 t = sh[0][g];
    for (int w = 1; w < 16; ++w) t += sh[w][g];
    *(f32x4*)(y + (long long)b * C + c) = f32x4{t[0] / (float)HW, t[1] / (float)HW, t[2] / (float)HW, t[3] / (float)HW};
  }
}
__global__ void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int lddx, int B, int HW, int C, int acc) {
  const int cpt = C >> 2;
  const long long total = (long long)B * HW * cpt;
  const float inv = 1.f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i / cpt;
    const int c = (int)(i - p * cpt) * 4;
    const long long b = p / HW;
    const f32x4 v = *(const f32x4*)(dy + b * C + c) * inv;
    f32x4* d = (f32x4*)(dx + p * lddx + c);
    *d = acc ? (*d + v) : v;
  }
}

// ------------------------------------------------------------------ softmax over pixels (per batch, per class column)
// [B][N][ld] logits, softmax along N for each of the K (<= 32) columns.  Three launches, all multi-block:
//   partial: per (batch, chunk of N) online (max, sum exp) per column -> ws[b][chunk][2][32]
//   (the merge of the chunk partials is recomputed by every block of the apply pass: tiny, L2 resident)
//   apply  : y = exp(x - m) / s
constexpr int SP_CHUNK = 512;  // pixels per block
__device__ __forceinline__ void sp_merge(const float* __restrict__ ws, int nch, int col, float& m, float& s) {
  m = -INFINITY;
  s = 0.f;
  for (int k = 0; k < nch; ++k) {
    const float mk = ws[(k * 2) * 32 + col], sk = ws[(k * 2 + 1) * 32 + col];
    const float mn = fmaxf(m, mk);
    s = s * expf(m - mn) + sk * expf(mk - mn);
    m = mn;
  }
}
__global__ __launch_bounds__(256) void softmax_spatial_partial_kernel(const float* __restrict__ x, float* __restrict__ ws, int N, int K, int ld,
                                                                      int nch) {
  const int b = blockIdx.y, ch = blockIdx.x;
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const float* xb = x + (long long)b * N * ld;
  const int n0 = ch * SP_CHUNK, n1 = min(n0 + SP_CHUNK, N);
  __shared__ float shm[8][33], shs[8][33];
  float m = -INFINITY, s = 0.f;
  if (col < K) {
    for (int n = n0 + rl; n < n1; n += 8) m = fmaxf(m, xb[(long long)n * ld + col]);
    for (int n = n0 + rl; n < n1; n += 8) s += expf(xb[(long long)n * ld + col] - m);
  }
  shm[rl][col] = m;
  shs[rl][col] = s;
  __syncthreads();
  if (rl == 0) {
    float M = shm[0][col];
#pragma unroll
    for (int k = 1; k < 8; ++k) M = fmaxf(M, shm[k][col]);
    float S = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) S += shm[k][col] == -INFINITY ? 0.f : shs[k][col] * expf(shm[k][col] - M);
    float* o = ws + ((long long)b * nch + ch) * 64;
    o[col] = M;
    o[32 + col] = S;
  }
}
__global__ __launch_bounds__(256) void softmax_spatial_apply_kernel(const float* __restrict__ x, const float* __restrict__ ws, float* __restrict__ y,
                                                                    int N, int K, int ld, int nch) {
  const int b = blockIdx.y, ch = blockIdx.x;
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  float m = 0.f, s = 1.f;
  if (col < K) sp_merge(ws + (long long)b * nch * 64, nch, col, m, s);
  const float inv = 1.f / s;
  const float* xb = x + (long long)b * N * ld;
  float* yb = y + (long long)b * N * ld;
  const int n0 = ch * SP_CHUNK, n1 = min(n0 + SP_CHUNK, N);
  for (int c0 = 0; c0 < ld; c0 += 32) {
    const int c = c0 + col;
    if (c >= ld) continue;
    for (int n = n0 + rl; n < n1; n += 8) yb[(long long)n * ld + c] = (c < K) ? expf(xb[(long long)n * ld + c] - m) * inv : 0.f;
  }
}
// backward: dot[b][k] = sum_n dy*y (partials per chunk), dx = y * (dy - dot)
__global__ __launch_bounds__(256) void softmax_spatial_bwd_partial_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                          float* __restrict__ ws, int N, int K, int ld, int nch) {
  const int b = blockIdx.y, ch = blockIdx.x;
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const float* yb = y + (long long)b * N * ld;
  const float* gb = dy + (long long)b * N * ld;
  const int n0 = ch * SP_CHUNK, n1 = min(n0 + SP_CHUNK, N);
  __shared__ float sh[8][33];
  float s = 0.f;
  if (col < K)
    for (int n = n0 + rl; n < n1; n += 8) s += gb[(long long)n * ld + col] * yb[(long long)n * ld + col];
  sh[rl][col] = s;
  __syncthreads();
  if (rl == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sh[k][col];
    ws[((long long)b * nch + ch) * 32 + col] = t;
  }
}
__global__ __launch_bounds__(256) void softmax_spatial_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                        const float* __restrict__ ws, float* __restrict__ dx, int N, int K,
                                                                        int ld, int nch, int acc) {
  const int b = blockIdx.y, ch = blockIdx.x;
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  float dot = 0.f;
  if (col < K)
    for (int k = 0; k < nch; ++k) dot += ws[((long long)b * nch + k) * 32 + col];
  const float* yb = y + (long long)b * N * ld;
  const float* gb = dy + (long long)b * N * ld;
  float* ob = dx + (long long)b * N * ld;
  const int n0 = ch * SP_CHUNK, n1 = min(n0 + SP_CHUNK, N);
  for (int c0 = 0; c0 < ld; c0 += 32) {
    const int c = c0 + col;
    if (c >= ld) continue;
    for (int n = n0 + rl; n < n1; n += 8) {
      const long long o = (long long)n * ld + c;
      const float v = (c < K) ? yb[o] * (gb[o] - dot) : 0.f;
      ob[o] = (acc && c < K) ? ob[o] + v : v;
    }
  }
}

// ------------------------------------------------------------------ softmax over <= 64 columns per row (one wave = one row)
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int K,
                                                               int ld, float scale) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long r = wave0; r < rows; r += nw) {
    const float v = lane < K ? scale * x[r * ld + lane] : -INFINITY;
    const float m = wave_max(v);
    const float e = lane < K ? expf(v - m) : 0.f;
    const float s = wave_sum(e);
    if (lane < ld) y[r * ld + lane] = e / s;
  }
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                               float* __restrict__ dx, long long rows, int K, int ld, float scale) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long r = wave0; r < rows; r += nw) {
    const float yy = lane < K ? y[r * ld + lane] : 0.f;
    const float g = lane < K ? dy[r * ld + lane] : 0.f;
    const float s = wave_sum(yy * g);
    if (lane < ld) dx[r * ld + lane] = scale * yy * (g - s);
  }
}

}  // namespace

extern "C" int catseg_nchw3_to_nhwc4(const float* x, float* y, int B, int H, int W, catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && cs_aligned16(y), "nchw3_to_nhwc4: bad args");
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(grid_for(B * HW)), dim3(256), 0, (hipStream_t)stream, x, y, B, HW);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_stem_pack_weight(const float* w, float* packed, int O, catseg_stream_t stream) {
  hipLaunchKernelGGL(stem_pack_kernel, dim3((O * 224 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, packed, O);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_stem_unpack_grad(const float* packed_grad, float* dw, int O, catseg_stream_t stream) {
  hipLaunchKernelGGL(stem_unpack_kernel, dim3((O * 147 + 255) / 256), dim3(256), 0, (hipStream_t)stream, packed_grad, dw, O);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_axpy2d(const float* src, int lds, float* dst, int ldd, long long rows, int C, float alpha,
                             int accumulate, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && cs_aligned16(src) && cs_aligned16(dst), "axpy2d: C/ld multiples of 4, 16-B aligned");
  hipLaunchKernelGGL(axpy2d_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, C, alpha, accumulate);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_add_n_act_amax(const float* const* in, const int* ld, int n, float* out, int ldo, long long rows, int C, int relu,
                                     void* amax_record, catseg_stream_t stream);
extern "C" int catseg_add_n_act(const float* const* in, const int* ld, int n, float* out, int ldo, long long rows, int C, int relu,
                                catseg_stream_t stream) {
  return catseg_add_n_act_amax(in, ld, n, out, ldo, rows, C, relu, nullptr, stream);
}
// the same, and max|out| folded into amax_record[0] (may be null)
extern "C" int catseg_add_n_act_amax(const float* const* in, const int* ld, int n, float* out, int ldo, long long rows, int C, int relu,
                                     void* amax_record, catseg_stream_t stream) {
  CS_REQUIRE(n >= 1 && n <= 4 && rows > 0 && C > 0 && C % 4 == 0 && ldo % 4 == 0 && cs_aligned16(out), "add_n: bad args");
  AddArgs a;
  a.n = n;
  for (int i = 0; i < 4; ++i) {
    a.in[i] = i < n ? in[i] : nullptr;
    a.ld[i] = i < n ? ld[i] : 0;
    if (i < n) CS_REQUIRE(cs_aligned16(in[i]) && ld[i] % 4 == 0, "add_n: input %d misaligned", i);
  }
  hipLaunchKernelGGL(add_n_act_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, out, ldo, rows, C, relu,
                     (unsigned*)amax_record);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
// catseg_add_n_act_amax that also writes the planes of out (exponent from the SUM of the terms' amax records: term_records[i] = the amax
// record of in[i], each the max over its 16 slots)
extern "C" int catseg_add_n_act_planes(const float* const* in, const int* ld, const void* const* term_records, int n, float* out, int ldo,
                                       void* out_planes, long long rows, int C, int relu, void* out_record, catseg_stream_t stream) {
  CS_REQUIRE(n >= 1 && n <= 4 && rows > 0 && C > 0 && C % 8 == 0 && ldo % 4 == 0 && cs_aligned16(out) && cs_aligned16(out_planes) && out_planes &&
                 out_record && term_records, "add_n (planes): bad args");
  AddArgs a;
  AddRecs rr;
  a.n = n;
  for (int i = 0; i < 4; ++i) {
    a.in[i] = i < n ? in[i] : nullptr;
    a.ld[i] = i < n ? ld[i] : 0;
    rr.rec[i] = i < n ? (const unsigned*)term_records[i] : nullptr;
    if (i < n) CS_REQUIRE(cs_aligned16(in[i]) && ld[i] % 4 == 0 && term_records[i], "add_n (planes): input %d misaligned / without a record", i);
  }
  const long long tiles = ((rows + 127) / 128) * ((C / 8 + 7) / 8);
  hipLaunchKernelGGL(add_n_act_planes_kernel, dim3((int)(tiles > 8192 ? 8192 : tiles)), dim3(256), 0, (hipStream_t)stream, a, rr, out, ldo,
                     (unsigned char*)out_planes, rows, C, relu, (unsigned*)out_record);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
namespace {
// out[b][i] (+)= sum_s slabs[b][s][i], s in a fixed order (four independent chains, combined the same way every time: deterministic)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const f32x4* __restrict__ slabs, f32x4* __restrict__ out, long long n4, int splits,
                                                        int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4* p = slabs + (long long)blockIdx.y * splits * n4 + i;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int k = 0;
  for (; k + 3 < splits; k += 4) {
    s0 += p[(long long)k * n4];
    s1 += p[(long long)(k + 1) * n4];
    s2 += p[(long long)(k + 2) * n4];
    s3 += p[(long long)(k + 3) * n4];
  }
  for (; k < splits; ++k) s0 += p[(long long)k * n4];
  f32x4 r = (s0 + s1) + (s2 + s3);
  f32x4* o = out + (long long)blockIdx.y * n4 + i;
  if (accumulate) r += *o;
  *o = r;
}
}  // namespace

// second stage of a split reduction (the K-split batched GEMMs of the OCR head: ops.gemm_tn_split): slabs [batch][splits][n] -> out [batch][n]
extern "C" int catseg_sum_slabs(const float* slabs, float* out, long long n, int splits, int batch, int accumulate, catseg_stream_t stream) {
  CS_REQUIRE(slabs && out && n > 0 && n % 4 == 0 && splits > 0 && batch > 0 && cs_aligned16(slabs) && cs_aligned16(out), "sum_slabs: bad args");
  const long long n4 = n / 4;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((n4 + 255) / 256), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const f32x4*)slabs,
                     (f32x4*)out, n4, splits, accumulate);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_relu_bwd(const float* dz, int lddz, const float* z, int ldz, float* g, int ldg, long long rows, int C,
                               catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && lddz % 4 == 0 && ldz % 4 == 0 && ldg % 4 == 0 && cs_aligned16(dz) && cs_aligned16(z) && cs_aligned16(g),
             "relu_bwd: bad args");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, dz, lddz, z, ldz, g, ldg, rows, C);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_weight_pad_cin(const float* w, float* out, int O, int taps, int cin, int cpad, int unpad, catseg_stream_t stream) {
  CS_REQUIRE(O > 0 && taps > 0 && cin > 0 && cpad >= cin, "weight_pad: bad args");
  const int n = O * taps * (unpad ? cin : cpad);
  hipLaunchKernelGGL(weight_pad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, n, cin, cpad, unpad);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_scale_by_device_scalar(float* x, long long n, const float* s, catseg_stream_t stream) {
  CS_REQUIRE(n > 0 && cs_aligned16(x) && s != nullptr, "scale: bad args");
  hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, n, s);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_maxpool3x3s2_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx, int B, int H, int W,
                                       int C, int Ho, int Wo, catseg_stream_t stream) {
  CS_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && Ho == (H + 1) / 2 && Wo == (W + 1) / 2 && cs_aligned16(x) && cs_aligned16(y) && cs_aligned16(idx), "maxpool fwd: bad args");
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long long)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, idx, B, H, W, C, Ho, Wo);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_maxpool3x3s2_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx, int B,
                                       int H, int W, int C, int Ho, int Wo, catseg_stream_t stream) {
  CS_REQUIRE(C % 4 == 0 && lddx % 4 == 0 && lddy % 4 == 0 && cs_aligned16(dx) && cs_aligned16(dy), "maxpool bwd: bad args");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, lddy, idx, dx, lddx, B, H, W, C, Ho, Wo);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, int Ho,
                                   int Wo, int align_corners, int accumulate, catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && ldx >= C && ldy >= C, "bilinear fwd: bad args");
  const float sh = resize_scale(H, Ho, align_corners), sw = resize_scale(W, Wo, align_corners);
  hipStream_t st = (hipStream_t)stream;
  if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && cs_aligned16(x) && cs_aligned16(y))
    hipLaunchKernelGGL(bilinear_fwd_kernel<2>, dim3(B * Ho), dim3(256), 0, st, x, ldx, y, ldy, B, H, W, C, Ho, Wo, align_corners, sh, sw, accumulate);
  else if (ldy == C && (Wo * C) % 4 == 0 && cs_aligned16(y)) {
    const size_t lds = (size_t)2 * W * ldx * 4;
    if (ldx % 4 == 0 && cs_aligned16(x) && lds <= 64 * 1024 && Ho >= 2 * H)      // (an upsample: every source row serves several output rows)
      hipLaunchKernelGGL(bilinear_fwd_lds_kernel, dim3(B * Ho), dim3(256), lds, st, x, ldx, y, B, H, W, C, Ho, Wo, align_corners, sh, sw, accumulate);
    else
      hipLaunchKernelGGL(bilinear_fwd_kernel<1>, dim3(B * Ho), dim3(256), 0, st, x, ldx, y, ldy, B, H, W, C, Ho, Wo, align_corners, sh, sw, accumulate);
  }
  else
    hipLaunchKernelGGL(bilinear_fwd_kernel<0>, dim3(B * Ho), dim3(256), 0, st, x, ldx, y, ldy, B, H, W, C, Ho, Wo, align_corners, sh, sw, accumulate);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int C,
                                   int Ho, int Wo, int align_corners, int zero_to, int accumulate, void* workspace,
                                   size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && lddx >= C && lddy >= C && zero_to <= lddx, "bilinear bwd: bad args");
  const size_t need = (size_t)B * H * Wo * C * 4;
  if (workspace_bytes < need || !workspace) {
    catseg_set_error("bilinear bwd: workspace %zu < %zu", workspace_bytes, need);
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  // ONE launch with the intermediate row in LDS where the layout allows 16-byte loads (bilinear_bwd_fused_kernel; bit-identical to the two passes)
  if (catseg_g_bilinear_bwd_fused) {
    const int mode = (lddy == C && (Wo * C) % 4 == 0 && cs_aligned16(dy)) ? 1 : ((C % 4 == 0 && lddy % 4 == 0 && cs_aligned16(dy)) ? 2 : 0);
    const float shf = resize_scale(H, Ho, align_corners);
    // (candidate rows of an input row: ~2 / sh + 6; the kernel lists at most BIL_MAXROWS of them)
    const bool rows_fit = shf > 0.f ? (2.0f / shf + 8.0f <= (float)BIL_MAXROWS) : (Ho <= BIL_MAXROWS);
    if (mode != 0 && rows_fit) {
      BilinearBwdFused a;
      a.dy = dy; a.lddy = lddy; a.dx = dx; a.lddx = lddx;
      a.B = B; a.H = H; a.W = W; a.C = C; a.Ho = Ho; a.Wo = Wo; a.align = align_corners;
      a.sh = resize_scale(H, Ho, align_corners); a.sw = resize_scale(W, Wo, align_corners);
      a.zero_to = zero_to; a.acc = accumulate;
      a.CC = mode == 1 ? C : (C < 96 ? C : 96);
      a.nchunk = mode == 1 ? 1 : (C + a.CC - 1) / a.CC;
      constexpr int CAP = 10240;                      // 40 KB of LDS per block: three blocks per CU
      // the longest segment that fits, halved while the grid is shorter than ~4 blocks per CU
      int seg = W, need = 0;
      while (seg >= 1 && (need = bil_fused_lds_floats(W, Wo, C, a.CC, mode, align_corners, a.sw, seg, CAP)) == 0) seg = seg > 1 ? (seg + 1) / 2 : 0;
      while (seg > 8 && (long long)B * H * ((W + seg - 1) / seg) * a.nchunk < 1024) {
        seg = (seg + 1) / 2;
        need = bil_fused_lds_floats(W, Wo, C, a.CC, mode, align_corners, a.sw, seg, CAP);
      }
      if (seg >= 1 && need > 0) {
        a.seg = seg; a.nseg = (W + seg - 1) / seg;
        const dim3 grid(B * H, a.nseg * a.nchunk);
        const size_t lds = (size_t)need * 4;
        if (mode == 1)
          hipLaunchKernelGGL(bilinear_bwd_fused_kernel<1>, grid, dim3(256), lds, st, a);
        else
          hipLaunchKernelGGL(bilinear_bwd_fused_kernel<2>, grid, dim3(256), lds, st, a);
        CS_LAUNCH_CHECK();
        return CATSEG_OK;
      }
    }
  }
  // few, long input rows at low resolution: split each row's sweep so that the grid still fills the chip
  const int ysplit_r = (int)((2048 + (long long)B * H - 1) / ((long long)B * H)) < (Wo * C + 255) / 256
                           ? (int)((2048 + (long long)B * H - 1) / ((long long)B * H)) : (Wo * C + 255) / 256;
  const int ysplit_c = (int)((2048 + (long long)B * H - 1) / ((long long)B * H)) < (W * C + 255) / 256
                           ? (int)((2048 + (long long)B * H - 1) / ((long long)B * H)) : (W * C + 255) / 256;
  const dim3 grid_r(B * H, ysplit_r < 1 ? 1 : ysplit_r);
  const float shr = resize_scale(H, Ho, align_corners);
  if (lddy == C && (Wo * C) % 4 == 0 && cs_aligned16(dy) && cs_aligned16(workspace))
    hipLaunchKernelGGL(bilinear_bwd_rows_kernel<1>, grid_r, dim3(256), 0, st, dy, lddy, (float*)workspace, B, H, C, Ho, Wo, align_corners, shr);
  else if (C % 4 == 0 && lddy % 4 == 0 && cs_aligned16(dy) && cs_aligned16(workspace))
    hipLaunchKernelGGL(bilinear_bwd_rows_kernel<2>, grid_r, dim3(256), 0, st, dy, lddy, (float*)workspace, B, H, C, Ho, Wo, align_corners, shr);
  else
    hipLaunchKernelGGL(bilinear_bwd_rows_kernel<0>, grid_r, dim3(256), 0, st, dy, lddy, (float*)workspace, B, H, C, Ho, Wo, align_corners, shr);
  hipLaunchKernelGGL(bilinear_bwd_cols_kernel, dim3(B * H, ysplit_c < 1 ? 1 : ysplit_c), dim3(256), 0, st, (const float*)workspace, dx, lddx, W, C, Wo, align_corners,
                     resize_scale(W, Wo, align_corners), zero_to, accumulate);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_global_avgpool_fwd(const float* x, int ldx, float* y, int B, int HW, int C, catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && HW > 0 && C > 0, "gap fwd: bad args");
  if (C % 4 == 0 && ldx % 4 == 0 && cs_aligned16(x) && cs_aligned16(y) && HW >= 256)
    hipLaunchKernelGGL(gap_fwd4_kernel, dim3((C + 63) / 64, B), dim3(1024), 0, (hipStream_t)stream, x, ldx, y, HW, C);
  else
    hipLaunchKernelGGL(gap_fwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, ldx, y, HW, C);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_global_avgpool_bwd(const float* dy, float* dx, int lddx, int B, int HW, int C, int accumulate,
                                         catseg_stream_t stream) {
  CS_REQUIRE(C % 4 == 0 && lddx % 4 == 0 && cs_aligned16(dy) && cs_aligned16(dx), "gap bwd: bad args");
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for((long long)B * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dx, lddx, B, HW, C, accumulate);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" size_t catseg_softmax_spatial_workspace(int B, int N) { return (size_t)B * ((N + SP_CHUNK - 1) / SP_CHUNK) * 64 * 4 + 256; }
extern "C" int catseg_softmax_spatial_fwd(const float* x, float* y, int B, int N, int K, int ld, void* workspace, size_t workspace_bytes,
                                          catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && N > 0 && K > 0 && K <= 32 && ld >= K && workspace != nullptr, "softmax spatial: bad args (K <= 32, workspace required)");
  const int nch = (N + SP_CHUNK - 1) / SP_CHUNK;
  CS_REQUIRE(workspace_bytes >= (size_t)B * nch * 64 * 4, "softmax spatial: workspace too small");
  hipLaunchKernelGGL(softmax_spatial_partial_kernel, dim3(nch, B), dim3(256), 0, (hipStream_t)stream, x, (float*)workspace, N, K, ld, nch);
  hipLaunchKernelGGL(softmax_spatial_apply_kernel, dim3(nch, B), dim3(256), 0, (hipStream_t)stream, x, (const float*)workspace, y, N, K, ld, nch);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_softmax_spatial_bwd(const float* y, const float* dy, float* dx, int B, int N, int K, int ld,
                                          int accumulate, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(B > 0 && N > 0 && K > 0 && K <= 32 && ld >= K && workspace != nullptr, "softmax spatial bwd: bad args (K <= 32, workspace required)");
  const int nch = (N + SP_CHUNK - 1) / SP_CHUNK;
  CS_REQUIRE(workspace_bytes >= (size_t)B * nch * 64 * 4, "softmax spatial bwd: workspace too small");
  hipLaunchKernelGGL(softmax_spatial_bwd_partial_kernel, dim3(nch, B), dim3(256), 0, (hipStream_t)stream, y, dy, (float*)workspace, N, K, ld, nch);
  hipLaunchKernelGGL(softmax_spatial_bwd_apply_kernel, dim3(nch, B), dim3(256), 0, (hipStream_t)stream, y, dy, (const float*)workspace, dx, N, K, ld, nch, accumulate);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_softmax_rows_fwd(const float* x, float* y, long long rows, int K, int ld, float scale,
                                       catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && K > 0 && K <= 64 && ld <= 64 && ld >= K, "softmax rows: K, ld <= 64");
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, K, ld, scale);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
extern "C" int catseg_softmax_rows_bwd(const float* y, const float* dy, float* dx, long long rows, int K, int ld,
                                       float scale, catseg_stream_t stream) {
  CS_REQUIRE(rows > 0 && K > 0 && K <= 64 && ld <= 64 && ld >= K, "softmax rows bwd: K, ld <= 64");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, rows, K, ld, scale);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
