// fp16 x 2 OPERAND PLANES of an NHWC fp32 activation (round 4): what the direct 3x3 kernels of csrc/dconv3_pl.hip / dwgrad3_pl.hip stream
// into LDS by LDS-DMA, written by the PRODUCER of the tensor (BatchNorm apply, the HRNet fuse sum, BatchNorm backward) instead of being
// split from fp32 inside every consuming convolution.
//
//   xs = x * 2^e (exact),   h = fp16(xs),   l = fp16(xs - h),   xs = h + l + r,  |r| <= 2^-22 |xs|        (csrc/igemm_f16x2.hip)
//
// Layout: [plane h / l][channel group g = C / 8][pixel][8 channels]  (16 bytes per (plane, group, pixel)): one (plane, group) slab is a
// dense pixel-major array, so a wave's LDS-DMA instruction fetches 64 halo pixels of one channel group as a few contiguous runs, and
// lands them in the [group][halo pixel][16 B] image the MFMA fragment reads want -- no VALU work in the consumer.
//
// The exponent e is known BEFORE the producer's pass over the data, from a bound of max|x| (never from the data itself: that would be a
// second pass): see cs_plane_exponent and the callers.  A bound that is too LARGE by a factor 2^k only moves the threshold below which
// an element's l plane becomes subnormal (|x| < 2^(k-17) max|x|; absolute error <= 2^-39+k max|x|); a bound that is too SMALL would
// overflow fp16 -- every bound used is a proven upper bound.
//
// Per-tensor record (uint32[CS_AMAX_WORDS], 2 KB, zeroed by the host): words 32 s (s = 0 .. 15) = bits of max|x| as the producer saw
// it (cs_amax_commit), word 1 = e (int32) the planes were written with, word 2 = bits of the bound a finalize kernel left, word 3 = bits of
// the bound e was finally derived from.
#pragma once
#include "common.h"

#define CS_REC_EXP 1
#define CS_REC_BOUND 2      /* what a finalize kernel leaves (atomicMax over the channels) BEFORE the producing pass reads it */
#define CS_REC_FINAL 3      /* the bound the exponent was finally derived from (CS_REC_BOUND + max|residual| ...): written by the producing pass */

// prescale exponent for a tensor whose max|x| is at most `bound` (bits of a non-negative float): bound * 2^e in [2^14, 2^15)
__host__ __device__ inline int cs_plane_exponent(unsigned bound_bits) {
  const int ex = (int)((bound_bits >> 23) & 0xFF);
  if (ex == 0 || ex == 255) return 0;
  const int e = 14 - (ex - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

#ifdef __HIPCC__
typedef _Float16 cs_h8 __attribute__((ext_vector_type(8)));

// Block-level writer: a 256-thread block holds a tile of 128 rows x up to 8 channel groups (64 channels).  The (row, group) items of a tile
// are numbered row-major (item = row * ngroups + group): item i of a pass belongs to thread i % 256, so that the threads of a wave read
// whole 32-byte groups of consecutive channels of consecutive rows (coalesced) and NO lane idles when the tensor has 6 groups (48
// channels: 768 items = 3 full passes).  stage(): a thread hands in 8 consecutive channels of one row (already scaled by 2^e); flush(): after
// a block barrier the tile leaves as 1 KB runs (64 rows x 16 B per wave instruction).  LDS image [plane][group][129 rows][16 B] (the odd
// row count spreads the groups a quarter-wave writes over the banks).
struct CsPlaneTile {
  static constexpr int ROWS = 128, GROUPS = 8, GS = (ROWS + 1) * 16, PS = GROUPS * GS;
  static constexpr int BYTES = 2 * PS;
  __device__ static __forceinline__ void stage(unsigned char* sm, int row, int group, const float (&xs)[8]) {
    cs_h8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 hh = (_Float16)xs[j];
      h[j] = hh;
      l[j] = (_Float16)(xs[j] - (float)hh);
    }
    *(cs_h8*)(sm + group * GS + row * 16) = h;
    *(cs_h8*)(sm + PS + group * GS + row * 16) = l;
  }
  // rows [row0, row0 + nrows) x groups [g0, g0 + ngroups) of a tensor with P pixels and NG = C / 8 groups
  __device__ static __forceinline__ void flush(const unsigned char* sm, unsigned char* planes, long long P, int NG, long long row0, int nrows,
                                               int g0, int ngroups) {
    for (int i = threadIdx.x; i < 2 * GROUPS * ROWS; i += blockDim.x) {
      const int row = i & (ROWS - 1), g = (i >> 7) & (GROUPS - 1), p = i >> 10;
      if (row < nrows && g < ngroups)
        *(cs_h8*)(planes + (((long long)p * NG + g0 + g) * P + row0 + row) * 16) = *(const cs_h8*)(sm + p * PS + g * GS + row * 16);
    }
  }
};
// tiles of a [rows][C] tensor and the (row, group) of item i of tile t: CS_PLANE_TILES / cs_plane_item
__device__ __forceinline__ long long cs_plane_tiles(long long rows, int NG) { return ((rows + CsPlaneTile::ROWS - 1) / CsPlaneTile::ROWS) * ((NG + 7) >> 3); }
#endif
