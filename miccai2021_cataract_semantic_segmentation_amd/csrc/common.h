// Shared helpers for the gfx950 kernels of libcatseg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "catseg.h"
#include "catseg_debug.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void catseg_set_error(const char* fmt, ...);

#define CS_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      catseg_set_error(__VA_ARGS__);          \
      return CATSEG_EINVAL;                   \
    }                                         \
  } while (0)

#define CS_LAUNCH_CHECK()                                                        \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      catseg_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,            \
                       hipGetErrorString(e__));                                  \
      return CATSEG_EHIP;                                                        \
    }                                                                            \
  } while (0)

#ifdef __HIPCC__
// per-tensor amax record of an ACTIVATION: CS_AMAX_SLOTS words 128 bytes apart (2 KB); a producing kernel folds the bits of its
// blocks' max |value| into slot blockIdx.x % CS_AMAX_SLOTS -- block reduction, then ONE fire-and-forget atomicMax per block, spread
// over 16 cache lines (a single word made every wave of a 8192-block launch queue on one L2 line: +18 ms per training step); the
// consumer takes the max over the slots.  Max is order-independent: deterministic.  Zeroed once per step by the host.
#define CS_AMAX_SLOTS 16
#define CS_AMAX_STRIDE 32        // words between slots
#define CS_AMAX_WORDS (CS_AMAX_SLOTS * CS_AMAX_STRIDE)
__device__ __forceinline__ void cs_amax_commit(unsigned m, unsigned* rec) {   // called by every thread of the block
  __shared__ unsigned cs_amax_sm[16];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) cs_amax_sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) m = max(m, cs_amax_sm[w]);
    if (m) atomicMax(rec + (blockIdx.x % CS_AMAX_SLOTS) * CS_AMAX_STRIDE, m);
  }
}
// the same into ONE word (small launches: the per-layer weight records)
__device__ __forceinline__ void cs_amax_commit1(unsigned m, unsigned* word) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(word, m);
}
// consumer side: max over the slots (every lane of the wave gets it)
__device__ __forceinline__ unsigned cs_amax_read(const unsigned* rec) {
  unsigned m = rec[(threadIdx.x & (CS_AMAX_SLOTS - 1)) * CS_AMAX_STRIDE];
#pragma unroll
  for (int o = CS_AMAX_SLOTS / 2; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  return m;
}
__device__ __forceinline__ unsigned cs_abs_bits4(const float __attribute__((ext_vector_type(4))) v) {
  return max(max(__float_as_uint(v[0]) & 0x7FFFFFFFu, __float_as_uint(v[1]) & 0x7FFFFFFFu),
             max(__float_as_uint(v[2]) & 0x7FFFFFFFu, __float_as_uint(v[3]) & 0x7FFFFFFFu));
}
#endif
#ifdef __HIPCC__
// Row-major walk over the float4 quads of a [rows][C] tensor WITHOUT a division in the loop (round 4).  The grid-stride form
// `r = i / cpt` with a 64-bit i cost ~80 VALU instructions per 16 bytes moved: the elementwise kernels were instruction-bound at ~0.5 of
// the HBM rate.  Here thread g of the launch visits quads g, g + S, g + 2 S, ... with S = (threads of the launch rounded down to a
// multiple of cpt): its channel quad never changes and its row advances by S / cpt -- one division per THREAD.  Consecutive threads still
// touch consecutive quads (coalesced).  Returns false for the (< cpt) threads past S; they must still take part in block-wide epilogues.
__device__ __forceinline__ bool cs_quad_walk(int cpt, long long& r, int& c, long long& rstep) {
  const long long T = (long long)gridDim.x * blockDim.x;
  rstep = T / cpt;
  const long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  r = g / cpt;
  c = (int)(g - r * cpt) * 4;
  return rstep > 0 && g < rstep * cpt;
}
#define CS_QUAD_LOOP(rows, cpt, r, c)                                                             \
  long long r, r##_step; int c;                                                                  \
  if (cs_quad_walk(cpt, r, c, r##_step))                                                         \
    for (; r < (rows); r += r##_step)
#endif
static inline bool cs_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline size_t cs_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// 64-wide wavefront reductions (CDNA: wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}


// ---- BatchNorm batch statistics in the convolution epilogue --------------------------------------------------------------
// Per (M-tile, channel) partials (K = tile mean, s1 = sum(v - K), s2 = sum((v - K)^2)) computed from the accumulator tile
// while it is still in registers: the separate statistics pass over the convolution output (one full HBM read of y)
// disappears; bn_finalize merges the tiles' partials exactly as it merges the row-block partials of bn_partial_kernel
// (fp64 Chan merge, fixed order: deterministic).
// A lane owns NS column slots (tile-local column col[s]) with R values each; lanes l ^ 32 (and l ^ 16 for the 16x16 MFMA map)
// hold other rows of the same columns, as do the WM waves stacked along M.  Must be called by every thread of the block.
//   lds: >= 2 * WM * tile_cols floats (the K-loop's buffers; the caller has synchronised the block after its last LDS read)
//   part: this tile's 3 x C partial rows (K, s1, s2), indexed by global column gcol0 + col
template <int NS, int R, int WM, bool X16, class Val, class Valid>
__device__ __forceinline__ void cs_tile_bn_partials(float* lds, int tile_cols, const int (&col)[NS], bool writer_lane, int wave_m, int nvalid,
                                                    Val val, Valid valid, float* part, int C, int gcol0) {
  float s[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) a += valid(i) ? val(j, i) : 0.f;
    a += __shfl_xor(a, 32, 64);
    if (X16) a += __shfl_xor(a, 16, 64);
    s[j] = a;
  }
  if (writer_lane) {
#pragma unroll
    for (int j = 0; j < NS; ++j) lds[wave_m * tile_cols + col[j]] = s[j];
  }
  __syncthreads();
  float mean[NS];
  const float inv = 1.f / (float)nvalid;
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) t += lds[w * tile_cols + col[j]];
    mean[j] = t * inv;
  }
  __syncthreads();
  float d1[NS], d2[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const float d = valid(i) ? val(j, i) - mean[j] : 0.f;
      a += d;
      b += d * d;
    }
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if (X16) {
      a += __shfl_xor(a, 16, 64);
      b += __shfl_xor(b, 16, 64);
    }
    d1[j] = a;
    d2[j] = b;
  }
  if (writer_lane) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      lds[wave_m * tile_cols + col[j]] = d1[j];
      lds[(WM + wave_m) * tile_cols + col[j]] = d2[j];
    }
  }
  __syncthreads();
  if (writer_lane && wave_m == 0) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int gc = gcol0 + col[j];
      if (gc < C) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          a += lds[w * tile_cols + col[j]];
          b += lds[(WM + w) * tile_cols + col[j]];
        }
        part[gc] = mean[j];
        part[C + gc] = a;
        part[2 * C + gc] = b;
      }
    }
  }
}
