// Lovasz-Softmax (losses/LovaszSoftmax.py:19-120 of the reference), cross entropy and the
// confusion matrix, as batched-over-classes HBM-bound kernels.
//
// Lovasz pipeline for logits [P][K], labels [P] (all classes in one launch, blockIdx.y = class,
// classes without a foreground pixel exit at once, no host round trip):
//   label_hist      : fg count per class, number of present classes
//   prep            : softmax over K per pixel; key[c][p] = 0x3F800000 - bits(|fg - p_c|) (ascending key
//                     = descending error, 30 significant bits), val[c][p] = p | fg << 31
//   4 x (upsweep, scan, downsweep): stable LSD radix sort, 8-bit digits; ranking inside a wave by
//                     ballot match (64-wide), wave->block->grid offsets through LDS, tiles written in digit order from LDS
//   fg block sums + scan, then `grad`: inclusive fg count F_i -> Jaccard gradient in fp32 exactly as
//                     lovasz_grad() computes it, loss partials, scatter of d loss / d prob to [c][pixel]
//   finalize        : loss = mean over present classes
//   backward        : softmax backward per pixel -> dlogits [P][K]
#include "common.h"

namespace {

constexpr int RBITS = 8;          // 4 passes of 8 bits over the 30-bit keys (256 digits: 16-element runs per 4096-key tile)
constexpr int RADIX = 1 << RBITS;
constexpr int SORT_PASSES = 4;
constexpr int SORT_BLOCKS = 128;  // blocks per class in upsweep / downsweep
constexpr int TILE = 4096;        // keys per tile (256 threads x 16)
constexpr int PIX = 256;          // pixels per block in the per-pixel kernels
constexpr int MAXK = 64;

struct LvWs {
  uint32_t* counts;   // [MAXK] fg count per class, [MAXK] = number of present classes
  uint32_t* keys[2];  // [K][P]
  uint32_t* vals[2];
  uint32_t* hist;     // [K][RADIX][SORT_BLOCKS]
  uint32_t* fgsum;    // [K][ntiles]
  float* lpart;       // [K][ntiles]
  float* dprob;       // [K][P]
  double* lossd;      // [1]
  uint32_t* minfg;    // [MAXK] bits of the smallest foreground error per class
  uint32_t* nact;     // [MAXK] length of the sorted sequence per class (P, or the active prefix when pruning)
  uint32_t* blkcnt;   // [K][nblk] active elements per 256-pixel block -> exclusive offsets
  unsigned long long* actmask;   // [P] bit c = class c keeps this pixel in its sort (active-set pruning)
};

size_t lv_layout(long long P, int K, char* base, LvWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += cs_align_up(bytes, 256);
    return p;
  };
  const long long ntiles = (P + TILE - 1) / TILE;
  LvWs t;
  t.counts = (uint32_t*)take((MAXK + 4) * 4);
  for (int i = 0; i < 2; ++i) {
    t.keys[i] = (uint32_t*)take((size_t)K * P * 4);
    t.vals[i] = (uint32_t*)take((size_t)K * P * 4);
  }
  t.hist = (uint32_t*)take((size_t)K * RADIX * SORT_BLOCKS * 4);
  t.fgsum = (uint32_t*)take((size_t)K * ntiles * 4);
  t.lpart = (float*)take((size_t)K * ntiles * 4);
  t.dprob = (float*)take((size_t)K * P * 4);
  t.lossd = (double*)take(64);
  t.minfg = (uint32_t*)take(MAXK * 4);
  t.nact = (uint32_t*)take(MAXK * 4);
  t.blkcnt = (uint32_t*)take((size_t)K * ((P + PIX - 1) / PIX) * 4);
  t.actmask = (unsigned long long*)take((size_t)P * 8);
  if (w) *w = t;
  return off;
}

// ---- stage a block of PIX pixels x K logits into LDS, row stride KS (odd).  The block's span starts at p0 * K floats with
// p0 a multiple of PIX, so it is 16-byte aligned whenever the tensor is: 16-byte global accesses (a scalar dword per lane
// moves ~1/4 of the bytes per request), one integer division per four elements instead of one each.
__device__ __forceinline__ void stage_rows(const float* __restrict__ logits, long long p0, int np, int K, int KS, float* sh) {
  const int n = np * K;
  const float* src = logits + p0 * K;
  int done = 0;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const int n4 = n >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    // Eight 16-byte loads per thread IN FLIGHT, then their LDS stores (a load -> LDS store loop waits for every load in turn -- the compiler
    // cannot move a global load over an LDS store through a generic pointer: 256 x 25 floats were 7 dependent HBM round trips per block and
    // the per-pixel kernels ran at ~2 TB/s)
    for (int base = 0; base < n4; base += 8 * blockDim.x) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i4 = base + u * blockDim.x + threadIdx.x;
        v[u] = i4 < n4 ? s4[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i4 = base + u * blockDim.x + threadIdx.x;
        if (i4 < n4) {
          const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          int r = (4 * i4) / K, c = 4 * i4 - r * K;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            sh[r * KS + c] = e[q];
            if (++c == K) { c = 0; ++r; }
          }
        }
      }
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / K, c = i - r * K;
    sh[r * KS + c] = src[i];
  }
}
// the reverse: rows of the LDS image to global (acc: added to what is there)
__device__ __forceinline__ void unstage_rows(float* __restrict__ out, long long p0, int np, int K, int KS, const float* sh, bool acc) {
  const int n = np * K;
  float* dst = out + p0 * K;
  int done = 0;
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const int n4 = n >> 2;
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i4 = threadIdx.x; i4 < n4; i4 += blockDim.x) {
      float e[4];
      int r = (4 * i4) / K, c = 4 * i4 - r * K;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        e[u] = sh[r * KS + c];
        if (++c == K) { c = 0; ++r; }
      }
      float4 v = make_float4(e[0], e[1], e[2], e[3]);
      if (acc) {
        const float4 o = d4[i4];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      d4[i4] = v;
    }
    done = n4 << 2;
  }
  for (int i = done + threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / K, c = i - r * K;
    const float v = sh[r * KS + c];
    dst[i] = acc ? dst[i] + v : v;
  }
}

__global__ void label_hist_kernel(const int64_t* __restrict__ labels, long long P, int K, uint32_t* __restrict__ counts) {
  __shared__ uint32_t h[MAXK];
  if (threadIdx.x < MAXK) h[threadIdx.x] = 0;
  __syncthreads();
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    const int64_t l = labels[i];
    if (l >= 0 && l < K) atomicAdd(&h[(int)l], 1u);
  }
  __syncthreads();
  if (threadIdx.x < K && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}
__global__ void present_kernel(int K, uint32_t* counts) {
  if (threadIdx.x == 0) {
    uint32_t n = 0;
    for (int c = 0; c < K; ++c) n += counts[c] != 0;
    counts[MAXK] = n;
  }
}

__global__ __launch_bounds__(PIX) void lv_prep_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long long P,
                                                      int K, const uint32_t* __restrict__ counts, uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
  extern __shared__ float sh[];
  const int KS = K | 1;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  stage_rows(logits, p0, np, K, KS, sh);
  __syncthreads();
  const int t = threadIdx.x;
  if (t >= np) return;
  float* row = sh + t * KS;
  float m = row[0];
  for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
  float s = 0.f;
  for (int c = 0; c < K; ++c) {
    const float e = expf(row[c] - m);
    row[c] = e;
    s += e;
  }
  const long long p = p0 + t;
  const int64_t lab = labels[p];
  for (int c = 0; c < K; ++c) {
    if (counts[c] == 0) continue;
    const float pr = row[c] / s;
    const uint32_t fg = (lab == c) ? 1u : 0u;
    const float err = fabsf((float)fg - pr);
    keys[(long long)c * P + p] = 0x3F800000u - __float_as_uint(err);
    vals[(long long)c * P + p] = (uint32_t)p | (fg << 31);
  }
}

// ---- active-set pruning ----------------------------------------------------------------------
// In the sorted order every element behind the LAST foreground pixel has Jaccard gradient exactly 0 (intersection is
// exhausted: J_i = J_{i-1} = 1), so it contributes +0 to the loss and 0 to d loss / d prob.  Only elements with
// err >= min over foreground pixels of err can precede that pixel: the rest never enters the sort.  The kept elements
// are compacted in pixel order (deterministic two-pass compaction), so the stable sort places them exactly where the
// full sort would: loss and gradient are bit-identical to the unpruned path.
__device__ __forceinline__ void softmax_row(float* row, int K, float& s) {
  float m = row[0];
  for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
  s = 0.f;
  for (int c = 0; c < K; ++c) {
    const float e = expf(row[c] - m);
    row[c] = e;
    s += e;
  }
}

__global__ __launch_bounds__(PIX) void lv_minfg_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long long P, int K,
                                                       uint32_t* __restrict__ minfg) {
  extern __shared__ float sh[];
  __shared__ uint32_t mn[MAXK];
  const int KS = K | 1;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  if (threadIdx.x < MAXK) mn[threadIdx.x] = 0x7F7F7F7Fu;
  stage_rows(logits, p0, np, K, KS, sh);
  __syncthreads();
  const int t = threadIdx.x;
  if (t < np) {
    const int64_t lab = labels[p0 + t];
    if (lab >= 0 && lab < K) {
      float* row = sh + t * KS;
      float s;
      softmax_row(row, K, s);
      const float err = fabsf(1.f - row[(int)lab] / s);
      atomicMin(&mn[(int)lab], __float_as_uint(err));  // err >= 0: unsigned order of the bits = order of the floats
    }
  }
  __syncthreads();
  if (threadIdx.x < K && mn[threadIdx.x] != 0x7F7F7F7Fu) atomicMin(&minfg[threadIdx.x], mn[threadIdx.x]);
}

// ONE pass over the logits decides, counts and stashes: per (class, 256-pixel block) the active elements are compacted INSIDE the block, in
// pixel order, into tmp[c][p0 + rank] = key | fg << 31 (keys have 30 significant bits), their number goes to blkcnt[c][block] and the pixel's
// class bits to actmask[p].  After the scan of the counts, lv_gather_kernel moves each block's run to its place in the class's sequence -- it
// reads the active elements only (round 4 ran this pass twice, count and write: a second read of the logits and a second softmax per pixel).
__global__ __launch_bounds__(PIX) void lv_compact1_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long long P, int K,
                                                          const uint32_t* __restrict__ counts, const uint32_t* __restrict__ minfg,
                                                          uint32_t* __restrict__ blkcnt, long long nblk, uint32_t* __restrict__ tmp,
                                                          unsigned long long* __restrict__ actmask) {
  extern __shared__ float sh[];
  __shared__ uint32_t wcnt[4][MAXK];
  const int KS = K | 1;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  stage_rows(logits, p0, np, K, KS, sh);
  __syncthreads();
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool live = t < np;
  float* row = sh + t * KS;
  float s = 1.f;
  int64_t lab = -1;
  if (live) {
    softmax_row(row, K, s);
    lab = labels[p0 + t];
  }
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  unsigned long long bits = 0;   // K <= 64
  for (int c = 0; c < K; ++c) {
    bool a = false;
    if (live && counts[c] != 0) {
      const uint32_t fg = (lab == c) ? 1u : 0u;
      const float err = fabsf((float)fg - row[c] / s);
      a = fg || __float_as_uint(err) >= minfg[c];
      if (a) row[c] = err;       // (kept for the write below: the element's key)
    }
    const unsigned long long bal = __ballot(a);
    if (lane == 0) wcnt[wave][c] = (uint32_t)__popcll(bal);
    if (a) bits |= 1ull << c;
  }
  if (live) actmask[p0 + t] = bits;
  __syncthreads();
  if (t < K) blkcnt[(long long)t * nblk + blockIdx.x] = wcnt[0][t] + wcnt[1][t] + wcnt[2][t] + wcnt[3][t];
  for (int c = 0; c < K; ++c) {
    const bool a = (bits >> c) & 1ull;
    const unsigned long long bal = __ballot(a);
    if (a) {
      uint32_t off = (uint32_t)__popcll(bal & lt_mask);
      for (int w2 = 0; w2 < wave; ++w2) off += wcnt[w2][c];
      const uint32_t fg = (lab == c) ? 1u : 0u;
      tmp[(long long)c * P + p0 + off] = (0x3F800000u - __float_as_uint(row[c])) | (fg << 31);
    }
  }
}

// blkoff = the exclusive scan of blkcnt (lv_blkscan_kernel); a block's run of class c is [blkoff[c][b], blkoff[c][b + 1]) (the last one ends at
// nact[c]).  val = the element's position in the class's compacted (pixel-ordered) sequence | fg << 31: d loss / d prob is stored in that
// compact domain (lv_grad_kernel scatters to it, lv_backward_kernel finds a pixel's slot from actmask and blkoff) -- no K x P plane is zeroed
// or read for the pruned elements.
__global__ __launch_bounds__(PIX) void lv_gather_kernel(const uint32_t* __restrict__ tmp, long long P, int K, const uint32_t* __restrict__ counts,
                                                        const uint32_t* __restrict__ blkoff, long long nblk, const uint32_t* __restrict__ nact,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  // every class's (offset, count) of this block in ONE round trip (thread = class), then wave w moves classes w, w + 4, ...: no load depends on
  // the previous class (a loop over the classes with its offsets fetched class by class took 167 us for 8 MB)
  __shared__ uint32_t soff[MAXK], scnt[MAXK];
  const long long p0 = (long long)blockIdx.x * PIX;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < K) {
    uint32_t off = 0, n = 0;
    if (counts[t] != 0) {
      off = blkoff[(long long)t * nblk + blockIdx.x];
      n = (blockIdx.x + 1 == nblk ? nact[t] : blkoff[(long long)t * nblk + blockIdx.x + 1]) - off;
    }
    soff[t] = off;
    scnt[t] = n;
  }
  __syncthreads();
  for (int c = wave; c < K; c += 4) {
    const uint32_t off = soff[c], n = scnt[c];
    for (uint32_t j = lane; j < n; j += 64) {
      const uint32_t v = tmp[(long long)c * P + p0 + j];
      keys[(long long)c * P + off + j] = v & 0x7FFFFFFFu;
      vals[(long long)c * P + off + j] = (off + j) | (v & 0x80000000u);
    }
  }
}

// exclusive scan of blkcnt[c][0..nblk) per class; the total is the length of the class's sort
__global__ __launch_bounds__(1024) void lv_blkscan_kernel(const uint32_t* __restrict__ counts, uint32_t* __restrict__ blkcnt, long long nblk,
                                                          uint32_t* __restrict__ nact) {
  const int c = blockIdx.x;
  if (counts[c] == 0) { if (threadIdx.x == 0) nact[c] = 0; return; }
  uint32_t* row = blkcnt + (long long)c * nblk;
  __shared__ uint32_t sh[1024];
  uint32_t carry = 0;
  for (long long b0 = 0; b0 < nblk; b0 += 1024) {
    const long long i = b0 + threadIdx.x;
    const uint32_t v = i < nblk ? row[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const uint32_t a = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < nblk) row[i] = carry + sh[threadIdx.x] - v;
    carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) nact[c] = carry;
}

__global__ void lv_fill_nact_kernel(uint32_t* nact, uint32_t P) {
  if (threadIdx.x < MAXK) nact[threadIdx.x] = P;
}

// ---- radix sort ---------------------------------------------------------------------------
__device__ __forceinline__ void block_range(long long P, long long& t0, long long& t1) {
  const long long ntiles = (P + TILE - 1) / TILE;
  const long long per = (ntiles + SORT_BLOCKS - 1) / SORT_BLOCKS;
  t0 = blockIdx.x * per;
  t1 = min(t0 + per, ntiles);
}

__global__ __launch_bounds__(256) void radix_upsweep_kernel(const uint32_t* __restrict__ keys, long long P, int shift,
                                                            const uint32_t* __restrict__ counts, const uint32_t* __restrict__ nact,
                                                            uint32_t* __restrict__ hist) {
  const int c = blockIdx.y;
  if (counts[c] == 0) return;
  const long long n = nact[c];  // elements of this class that take part in the sort (<= P)
  __shared__ uint32_t h[4][RADIX];   // one histogram per wave (less LDS-atomic contention on skewed digits)
  for (int i = threadIdx.x; i < 4 * RADIX; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  long long t0, t1;
  block_range(n, t0, t1);
  const uint32_t* k = keys + (long long)c * P;
  const long long e0 = t0 * TILE, e1 = min(t1 * TILE, n);
  uint32_t* hw = h[threadIdx.x >> 6];
  long long i = e0 + threadIdx.x;
  for (; i + 768 < e1; i += 1024) {   // four independent loads in flight per thread
    const uint32_t k0 = k[i], k1 = k[i + 256], k2 = k[i + 512], k3 = k[i + 768];
    atomicAdd(&hw[(k0 >> shift) & (RADIX - 1)], 1u);
    atomicAdd(&hw[(k1 >> shift) & (RADIX - 1)], 1u);
    atomicAdd(&hw[(k2 >> shift) & (RADIX - 1)], 1u);
    atomicAdd(&hw[(k3 >> shift) & (RADIX - 1)], 1u);
  }
  for (; i < e1; i += 256) atomicAdd(&hw[(k[i] >> shift) & (RADIX - 1)], 1u);
  __syncthreads();
  uint32_t* o = hist + (long long)c * RADIX * SORT_BLOCKS;
  for (int d = threadIdx.x; d < RADIX; d += 256) o[d * SORT_BLOCKS + blockIdx.x] = h[0][d] + h[1][d] + h[2][d] + h[3][d];
}

// exclusive scan of hist[c][d][b] in (d, b) order; one 1024-thread block per class: thread = (digit, quarter of the blocks)
__global__ __launch_bounds__(1024) void radix_scan_kernel(const uint32_t* __restrict__ counts, uint32_t* __restrict__ hist) {
  static_assert(RADIX * 4 == 1024 && SORT_BLOCKS % 4 == 0, "radix_scan_kernel: 4 threads per digit");
  constexpr int SEG = SORT_BLOCKS / 4;
  const int c = blockIdx.x;
  if (counts[c] == 0) return;
  uint32_t* row = hist + (long long)c * RADIX * SORT_BLOCKS + (long long)threadIdx.x * SEG;   // = [digit][quarter] in scan order
  uint32_t v[SEG];
  uint32_t tot = 0;
#pragma unroll
  for (int b = 0; b < SEG; ++b) {
    v[b] = row[b];
    tot += v[b];
  }
  __shared__ uint32_t sh[1024];
  sh[threadIdx.x] = tot;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t a = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += a;
    __syncthreads();
  }
  uint32_t run = sh[threadIdx.x] - tot;
#pragma unroll
  for (int b = 0; b < SEG; ++b) {
    row[b] = run;
    run += v[b];
  }
}

// One stable counting pass.  Per tile of 4096 (key, value) pairs: rank inside each wave by ballot match, per-wave -> per-tile digit
// offsets through LDS, then the tile is laid out in digit order IN LDS and written from there, so that consecutive lanes write
// consecutive addresses of a digit's run (the direct register -> global scatter of round 1 wrote 4-byte pieces to ~1000 runs per tile:
// 0.8 TB/s effective).
__global__ __launch_bounds__(256) void radix_downsweep_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                              uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                              long long P, int shift, const uint32_t* __restrict__ counts,
                                                              const uint32_t* __restrict__ nact, const uint32_t* __restrict__ hist) {
  static_assert(RADIX == 256, "radix_downsweep_kernel: one thread per digit");
  const int c = blockIdx.y;
  if (counts[c] == 0) return;
  const long long n = nact[c];
  __shared__ uint32_t whist[4][RADIX];  // per-wave digit counts of the current tile -> tile-local start of (wave, digit)
  __shared__ uint32_t running[RADIX];   // next global slot per digit for this block
  __shared__ uint32_t goff[RADIX];      // global slot of a digit's run minus its tile-local start
  __shared__ uint32_t wtot[4];
  __shared__ uint32_t skey[TILE], sval[TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* hb = hist + (long long)c * RADIX * SORT_BLOCKS;
  running[tid] = hb[tid * SORT_BLOCKS + blockIdx.x];
  const uint32_t* kin = keys_in + (long long)c * P;
  const uint32_t* vin = vals_in + (long long)c * P;
  uint32_t* kout = keys_out + (long long)c * P;
  uint32_t* vout = vals_out + (long long)c * P;
  long long t0, t1;
  block_range(n, t0, t1);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (long long t = t0; t < t1; ++t) {
    for (int i = tid; i < 4 * RADIX; i += 256) (&whist[0][0])[i] = 0;
    __syncthreads();
    const long long tile0 = t * TILE;
    const long long base = tile0 + wave * (TILE / 4);
    uint32_t key[16], val[16], rank[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long i = base + r * 64 + lane;
      const bool live = i < n;
      key[r] = live ? kin[i] : 0xFFFFFFFFu;
      val[r] = live ? vin[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long i = base + r * 64 + lane;
      const bool live = i < n;
      const uint32_t d = (key[r] >> shift) & (RADIX - 1);
      unsigned long long peers = __ballot(live);
#pragma unroll
      for (int b = 0; b < RBITS; ++b) {
        const unsigned long long bal = __ballot((d >> b) & 1);
        peers &= ((d >> b) & 1) ? bal : ~bal;
      }
      uint32_t old = 0;
      const int leader = __ffsll((long long)peers) - 1;
      if (live && lane == leader) {
        old = whist[wave][d];
        whist[wave][d] = old + __popcll(peers);
      }
      old = __shfl(old, leader < 0 ? 0 : leader, 64);
      rank[r] = old + __popcll(peers & lt_mask);
    }
    __syncthreads();
    {  // thread = digit: exclusive scan of the tile's digit counts, per-wave starts, global offset of the run
      const int d = tid;
      const uint32_t c0 = whist[0][d], c1 = whist[1][d], c2 = whist[2][d], c3 = whist[3][d];
      const uint32_t cnt = c0 + c1 + c2 + c3;
      uint32_t inc = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
      }
      if (lane == 63) wtot[wave] = inc;
      __syncthreads();
      uint32_t start = inc - cnt;
      for (int w = 0; w < wave; ++w) start += wtot[w];
      whist[0][d] = start;
      whist[1][d] = start + c0;
      whist[2][d] = start + c0 + c1;
      whist[3][d] = start + c0 + c1 + c2;
      const uint32_t g = running[d];
      goff[d] = g - start;
      running[d] = g + cnt;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long i = base + r * 64 + lane;
      if (i < n) {
        const uint32_t d = (key[r] >> shift) & (RADIX - 1);
        const uint32_t pos = whist[wave][d] + rank[r];
        skey[pos] = key[r];
        sval[pos] = val[r];
      }
    }
    __syncthreads();
    const int nv = (int)min((long long)TILE, n - tile0);
#pragma unroll 4
    for (int j = tid; j < nv; j += 256) {
      const uint32_t k = skey[j];
      const uint32_t dst = goff[(k >> shift) & (RADIX - 1)] + (uint32_t)j;
      kout[dst] = k;
      vout[dst] = sval[j];
    }
  }
}

// ---- Jaccard gradient over the sorted order -------------------------------------------------
__global__ __launch_bounds__(256) void lv_fgsum_kernel(const uint32_t* __restrict__ vals, long long P, const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ nact, uint32_t* __restrict__ fgsum, long long ntiles) {
  const int c = blockIdx.y;
  if (counts[c] == 0) return;
  const uint32_t* v = vals + (long long)c * P;
  const long long e0 = (long long)blockIdx.x * TILE, e1 = min(e0 + TILE, (long long)nact[c]);
  uint32_t n = 0;
  for (long long i = e0 + threadIdx.x; i < e1; i += 256) n += v[i] >> 31;
  n = (uint32_t)wave_sum((float)n);  // <= 4096, exact in fp32
  __shared__ uint32_t sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) fgsum[(long long)c * ntiles + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(1024) void lv_fgscan_kernel(const uint32_t* __restrict__ counts, uint32_t* __restrict__ fgsum, long long ntiles) {
  const int c = blockIdx.x;
  if (counts[c] == 0) return;
  uint32_t* row = fgsum + (long long)c * ntiles;
  __shared__ uint32_t sh[1024];
  uint32_t carry = 0;
  for (long long b0 = 0; b0 < ntiles; b0 += 1024) {
    const long long i = b0 + threadIdx.x;
    const uint32_t v = i < ntiles ? row[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const uint32_t a = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < ntiles) row[i] = carry + sh[threadIdx.x] - v;  // exclusive
    carry += sh[1023];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void lv_grad_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, long long P,
                                                      const uint32_t* __restrict__ counts, const uint32_t* __restrict__ nact,
                                                      const uint32_t* __restrict__ fgsum, long long ntiles, float* __restrict__ lpart,
                                                      float* __restrict__ dprob) {
  const int c = blockIdx.y;
  if (counts[c] == 0) return;
  const long long n = nact[c];
  if ((long long)blockIdx.x * TILE >= n) {  // behind the active prefix: nothing but an exact +0 to the loss
    if (threadIdx.x == 0) lpart[(long long)c * ntiles + blockIdx.x] = 0.f;
    return;
  }
  const uint32_t* k = keys + (long long)c * P;
  const uint32_t* v = vals + (long long)c * P;
  float* dp = dprob ? dprob + (long long)c * P : nullptr;
  const float G = (float)counts[c];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long e0 = (long long)blockIdx.x * TILE;
  // thread owns 16 consecutive elements: e0 + tid*16 .. +15
  uint32_t kk[16], vv[16];
  uint32_t nfg = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const long long i = e0 + tid * 16 + j;
    kk[j] = i < n ? k[i] : 0x3F800000u;
    vv[j] = i < n ? v[i] : 0u;
    nfg += vv[j] >> 31;
  }
  // exclusive scan of nfg over the block
  uint32_t incl = nfg;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t a = __shfl_up(incl, o, 64);
    if (lane >= o) incl += a;
  }
  __shared__ uint32_t wsum[4];
  __shared__ float lsum[4];
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t F = fgsum[(long long)c * ntiles + blockIdx.x] + incl - nfg;
  for (int w = 0; w < wave; ++w) F += wsum[w];
  float loss = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const long long i = e0 + tid * 16 + j;
    if (i < n) {
      const uint32_t f = vv[j] >> 31;
      // J_{i-1} from F (fg count before i), J_i after including i
      const float Fb = (float)F;
      const float jprev = (i == 0) ? 0.f : 1.f - (G - Fb) / (G + ((float)i - Fb));
      F += f;
      const float Fa = (float)F;
      const float jcur = 1.f - (G - Fa) / (G + ((float)(i + 1) - Fa));
      const float g = jcur - jprev;
      const float err = __uint_as_float(0x3F800000u - kk[j]);
      loss += err * g;
      if (dp) dp[vv[j] & 0x7FFFFFFFu] = (err == 0.f) ? 0.f : (f ? -g : g);
    }
  }
  loss = wave_sum(loss);
  if (lane == 0) lsum[wave] = loss;
  __syncthreads();
  if (tid == 0) lpart[(long long)c * ntiles + blockIdx.x] = lsum[0] + lsum[1] + lsum[2] + lsum[3];
}

__global__ __launch_bounds__(256) void lv_finalize_kernel(const uint32_t* __restrict__ counts, const float* __restrict__ lpart, long long ntiles,
                                                          int K, float weight, float* __restrict__ loss_out, int accumulate) {
  __shared__ double sh[256];
  double tot = 0;
  for (int c = 0; c < K; ++c) {
    if (counts[c] == 0) continue;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;      // four independent load chains (one chain of ~100 dependent loads per class was 25 - 57 us)
    long long i = threadIdx.x;
    for (; i + 768 < ntiles; i += 1024) {
      s0 += lpart[(long long)c * ntiles + i];
      s1 += lpart[(long long)c * ntiles + i + 256];
      s2 += lpart[(long long)c * ntiles + i + 512];
      s3 += lpart[(long long)c * ntiles + i + 768];
    }
    for (; i < ntiles; i += 256) s0 += lpart[(long long)c * ntiles + i];
    tot += (s0 + s1) + (s2 + s3);
  }
  sh[threadIdx.x] = tot;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint32_t n = counts[MAXK];
    const float l = n ? (float)(sh[0] / (double)n) * weight : 0.f;
    loss_out[0] = accumulate ? loss_out[0] + l : l;
  }
}

// actmask != nullptr (active-set pruning): d loss / d prob lives in the compact domain of lv_gather_kernel -- the slot of (pixel, class) is
// blkoff[c][block] + the number of active pixels of class c in front of this one inside the block (ballot ranks, as lv_compact1_kernel
// counted them); a pruned element's gradient is exactly 0 and is never stored.  actmask == nullptr: dprob is [K][P] by pixel.
__global__ __launch_bounds__(PIX) void lv_backward_kernel(const float* __restrict__ logits, long long P, int K, const uint32_t* __restrict__ counts,
                                                          const float* __restrict__ dprob, float weight, float* __restrict__ dlogits, int acc,
                                                          const float* __restrict__ upstream, const unsigned long long* __restrict__ actmask,
                                                          const uint32_t* __restrict__ blkoff, long long nblk) {
  extern __shared__ float sh[];
  __shared__ uint32_t wcnt[4][MAXK];
  const int KS = K | 1;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  stage_rows(logits, p0, np, K, KS, sh);
  const int t = threadIdx.x;
  unsigned long long bits = 0;
  if (actmask) {
    const int lane = t & 63, wave = t >> 6;
    if (t < np) bits = actmask[p0 + t];
    for (int c = 0; c < K; ++c) {
      const unsigned long long bal = __ballot((bits >> c) & 1ull);
      if (lane == 0) wcnt[wave][c] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (t < K) {        // thread = class: wcnt[w][c] <- first compact slot of wave w's pixels of class c
      const uint32_t b = blkoff[(long long)t * nblk + blockIdx.x], w0 = wcnt[0][t], w1 = wcnt[1][t], w2 = wcnt[2][t];
      wcnt[0][t] = b; wcnt[1][t] = b + w0; wcnt[2][t] = b + w0 + w1; wcnt[3][t] = b + w0 + w1 + w2;
    }
  }
  __syncthreads();
  if (t < np) {
    float* row = sh + t * KS;
    float m = row[0];
    for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
    float s = 0.f;
    for (int c = 0; c < K; ++c) {
      const float e = expf(row[c] - m);
      row[c] = e;
      s += e;
    }
    const uint32_t npres = counts[MAXK];
    const float w = npres ? weight / (float)npres : 0.f;
    const long long p = p0 + t;
    float* grow = sh + PIX * KS + t * KS;  // second LDS image: d loss / d prob of this pixel
    float dot = 0.f;
    for (int c0 = 0; c0 < K; c0 += 8) {   // eight independent class-plane loads in flight (absent classes: dprob is not written, g = 0)
      float gv[8];
      if (actmask) {
        const int lane = t & 63, wave = t >> 6;
        const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = c0 + u;
          gv[u] = 0.f;
          if (c < K) {
            // (every lane of the wave is here: np is a multiple of 64 except in the tensor's last block, whose idle lanes hold bits = 0 and
            //  are not inside this branch -- their ballot bit is 0 either way)
            const bool a = (bits >> c) & 1ull;
            const unsigned long long bal = __ballot(a);
            if (a) gv[u] = dprob[(long long)c * P + wcnt[wave][c] + (uint32_t)__popcll(bal & lt_mask)];
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = c0 + u;
          gv[u] = (c < K && counts[c]) ? dprob[(long long)c * P + p] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u;
        if (c < K) {
          const float pr = row[c] / s;
          const float g = gv[u] * w;
          row[c] = pr;
          grow[c] = g;
          dot += g * pr;
        }
      }
    }
    // upstream: d(total loss) / d(this loss), a device scalar (autograd's grad_output): the product the separate scaling pass used to form,
    // (d loss / d logit) * upstream, rounded the same way
    const float up = upstream ? *upstream : 1.f;
    for (int c = 0; c < K; ++c) {
      const float v = row[c] * (grow[c] - dot);
      row[c] = upstream ? v * up : v;
    }
  }
  __syncthreads();
  unstage_rows(dlogits, p0, np, K, KS, sh, acc != 0);
}

// ---- cross entropy ----------------------------------------------------------------------------
__global__ __launch_bounds__(PIX) void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long long P, int K,
                                                     long long ignore, float* __restrict__ part) {
  extern __shared__ float sh[];
  const int KS = K | 1;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  stage_rows(logits, p0, np, K, KS, sh);
  __syncthreads();
  const int t = threadIdx.x;
  float l = 0.f, cnt = 0.f;
  if (t < np) {
    const int64_t lab = labels[p0 + t];
    if (lab != ignore && lab >= 0 && lab < K) {
      const float* row = sh + t * KS;
      float m = row[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
      float s = 0.f;
      for (int c = 0; c < K; ++c) s += expf(row[c] - m);
      l = logf(s) + m - row[(int)lab];
      cnt = 1.f;
    }
  }
  l = wave_sum(l);
  cnt = wave_sum(cnt);
  __shared__ float r[8];
  if ((t & 63) == 0) { r[t >> 6] = l; r[4 + (t >> 6)] = cnt; }
  __syncthreads();
  if (t == 0) {
    part[2 * blockIdx.x] = r[0] + r[1] + r[2] + r[3];
    part[2 * blockIdx.x + 1] = r[4] + r[5] + r[6] + r[7];
  }
}
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ part, long long nb, float weight, float* __restrict__ loss_out,
                                                          float* __restrict__ inv_count) {
  __shared__ double s1[256], s2[256];
  double a = 0, b = 0;
  for (long long i = threadIdx.x; i < nb; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss_out[0] = (float)(s1[0] / s2[0]) * weight;  // 0/0 -> nan, as torch
    inv_count[0] = s2[0] > 0 ? (float)(1.0 / s2[0]) : 0.f;
  }
}
__global__ __launch_bounds__(PIX) void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long long P, int K,
                                                     long long ignore, float weight, const float* __restrict__ inv_count,
                                                     float* __restrict__ dlogits) {
  extern __shared__ float sh[];
  const int KS = K | 1;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  stage_rows(logits, p0, np, K, KS, sh);
  __syncthreads();
  const int t = threadIdx.x;
  if (t < np) {
    float* row = sh + t * KS;
    const int64_t lab = labels[p0 + t];
    if (lab != ignore && lab >= 0 && lab < K) {
      float m = row[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
      float s = 0.f;
      for (int c = 0; c < K; ++c) { const float e = expf(row[c] - m); row[c] = e; s += e; }
      const float w = weight * inv_count[0];
      for (int c = 0; c < K; ++c) row[c] = (row[c] / s - (c == (int)lab ? 1.f : 0.f)) * w;
    } else {
      for (int c = 0; c < K; ++c) row[c] = 0.f;
    }
  }
  __syncthreads();
  unstage_rows(dlogits, p0, np, K, KS, sh, false);
}

// ---- confusion matrix ---------------------------------------------------------------------------
__global__ __launch_bounds__(PIX) void confusion_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long long P, int K,
                                                        int32_t* __restrict__ cm) {
  extern __shared__ float sh[];
  const int KS = K | 1;
  int* hist = (int*)(sh + PIX * KS);
  for (int i = threadIdx.x; i < K * K; i += blockDim.x) hist[i] = 0;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  stage_rows(logits, p0, np, K, KS, sh);
  __syncthreads();
  const int t = threadIdx.x;
  if (t < np) {
    const float* row = sh + t * KS;
    int best = 0;
    float bv = row[0];
    for (int c = 1; c < K; ++c)
      if (row[c] > bv) { bv = row[c]; best = c; }
    const int64_t lab = labels[p0 + t];
    if (lab >= 0 && lab < K) atomicAdd(&hist[best * K + (int)lab], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K * K; i += blockDim.x)
    if (hist[i]) atomicAdd(&cm[i], hist[i]);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n4,
                            long long n, float lr, float b1, float b2, float eps, float bc1, float bc2s, float gscale,
                            const float* __restrict__ hyper) {
  if (hyper) {      // step-dependent scalars from device memory (a captured launch replays with new values): {lr, bc1, sqrt(bc2), grad scale}
    lr = hyper[0]; bc1 = hyper[1]; bc2s = hyper[2]; gscale = hyper[3];
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    if (e + 3 < n) {
      f32x4 gg = *(const f32x4*)(g + e) * gscale;
      f32x4 mm = *(f32x4*)(m + e) * b1 + gg * (1.f - b1);
      f32x4 vv = *(f32x4*)(v + e) * b2 + gg * gg * (1.f - b2);
      f32x4 pp = *(f32x4*)(p + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) pp[k] -= (lr / bc1) * (mm[k] / (sqrtf(vv[k]) / bc2s + eps));
      *(f32x4*)(m + e) = mm; *(f32x4*)(v + e) = vv; *(f32x4*)(p + e) = pp;
    } else {
      for (long long j = e; j < n; ++j) {
        const float gg = g[j] * gscale;
        const float mm = m[j] * b1 + gg * (1.f - b1);
        const float vv = v[j] * b2 + gg * gg * (1.f - b2);
        m[j] = mm; v[j] = vv;
        p[j] -= (lr / bc1) * (mm / (sqrtf(vv) / bc2s + eps));
      }
    }
  }
}

int g_prune = 1;

}  // namespace

extern "C" int catseg_debug_set_lovasz_prune(int on) {
  g_prune = on ? 1 : 0;
  return CATSEG_OK;
}

extern "C" size_t catseg_lovasz_workspace(long long P, int K) { return lv_layout(P, K, nullptr, nullptr); }

namespace {
int lovasz_run(const float* logits, const int64_t* labels, long long P, int K, float weight, float* loss_out, float* dlogits, int want_grad,
               int accumulate_dlogits, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(P > 0 && P < (1ll << 31) && K > 0 && K <= MAXK, "lovasz: need 0 < P < 2^31 and K <= %d", MAXK);
  const size_t need = lv_layout(P, K, nullptr, nullptr);
  if (workspace_bytes < need || !workspace) {
    catseg_set_error("lovasz: workspace %zu < %zu", workspace_bytes, need);
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  LvWs w;
  lv_layout(P, K, (char*)workspace, &w);
  const long long ntiles = (P + TILE - 1) / TILE;
  const int nb = (int)((P + PIX - 1) / PIX);
  const size_t shb = (size_t)PIX * (K | 1) * 4;
  if (hipMemsetAsync(w.counts, 0, (MAXK + 4) * 4, st) != hipSuccess) { catseg_set_error("lovasz: memset failed"); return CATSEG_EHIP; }
  hipLaunchKernelGGL(label_hist_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, st, labels, P, K, w.counts);
  hipLaunchKernelGGL(present_kernel, dim3(1), dim3(64), 0, st, K, w.counts);
  if (g_prune) {
    // active-set pruning: min foreground error per class -> count -> scan -> ordered compaction (3 passes over the logits)
    if (hipMemsetAsync(w.minfg, 0x7F, MAXK * 4, st) != hipSuccess) { catseg_set_error("lovasz: memset failed"); return CATSEG_EHIP; }
    hipLaunchKernelGGL(lv_minfg_kernel, dim3(nb), dim3(PIX), shb, st, logits, labels, P, K, w.minfg);
    hipLaunchKernelGGL(lv_compact1_kernel, dim3(nb), dim3(PIX), shb, st, logits, labels, P, K, (const uint32_t*)w.counts,
                       (const uint32_t*)w.minfg, w.blkcnt, (long long)nb, w.keys[1], w.actmask);
    hipLaunchKernelGGL(lv_blkscan_kernel, dim3(K), dim3(1024), 0, st, (const uint32_t*)w.counts, w.blkcnt, (long long)nb, w.nact);
    hipLaunchKernelGGL(lv_gather_kernel, dim3(nb), dim3(PIX), 0, st, (const uint32_t*)w.keys[1], P, K, (const uint32_t*)w.counts,
                       (const uint32_t*)w.blkcnt, (long long)nb, (const uint32_t*)w.nact, w.keys[0], w.vals[0]);
    // (d loss / d prob in the compact domain: every slot below nact[c] is written by lv_grad_kernel, nothing beyond it is read)
  } else {
    hipLaunchKernelGGL(lv_fill_nact_kernel, dim3(1), dim3(64), 0, st, w.nact, (uint32_t)P);
    hipLaunchKernelGGL(lv_prep_kernel, dim3(nb), dim3(PIX), shb, st, logits, labels, P, K, (const uint32_t*)w.counts, w.keys[0], w.vals[0]);
  }
  const uint32_t* nact = (const uint32_t*)w.nact;
  int cur = 0;
  for (int pass = 0; pass < SORT_PASSES; ++pass) {
    const int shift = pass * RBITS;
    hipLaunchKernelGGL(radix_upsweep_kernel, dim3(SORT_BLOCKS, K), dim3(256), 0, st, (const uint32_t*)w.keys[cur], P, shift, (const uint32_t*)w.counts, nact, w.hist);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(K), dim3(1024), 0, st, (const uint32_t*)w.counts, w.hist);
    hipLaunchKernelGGL(radix_downsweep_kernel, dim3(SORT_BLOCKS, K), dim3(256), 0, st, (const uint32_t*)w.keys[cur], (const uint32_t*)w.vals[cur],
                       w.keys[cur ^ 1], w.vals[cur ^ 1], P, shift, (const uint32_t*)w.counts, nact, (const uint32_t*)w.hist);
    cur ^= 1;
  }
  hipLaunchKernelGGL(lv_fgsum_kernel, dim3((unsigned)ntiles, K), dim3(256), 0, st, (const uint32_t*)w.vals[cur], P, (const uint32_t*)w.counts, nact, w.fgsum, ntiles);
  hipLaunchKernelGGL(lv_fgscan_kernel, dim3(K), dim3(1024), 0, st, (const uint32_t*)w.counts, w.fgsum, ntiles);
  hipLaunchKernelGGL(lv_grad_kernel, dim3((unsigned)ntiles, K), dim3(256), 0, st, (const uint32_t*)w.keys[cur], (const uint32_t*)w.vals[cur], P,
                     (const uint32_t*)w.counts, nact, (const uint32_t*)w.fgsum, ntiles, w.lpart, want_grad ? w.dprob : nullptr);
  hipLaunchKernelGGL(lv_finalize_kernel, dim3(1), dim3(256), 0, st, (const uint32_t*)w.counts, (const float*)w.lpart, ntiles, K, weight, loss_out, 0);
  if (dlogits)
    hipLaunchKernelGGL(lv_backward_kernel, dim3(nb), dim3(PIX), 2 * shb, st, logits, P, K, (const uint32_t*)w.counts, (const float*)w.dprob, weight, dlogits,
                       accumulate_dlogits, (const float*)nullptr, g_prune ? (const unsigned long long*)w.actmask : nullptr, (const uint32_t*)w.blkcnt,
                       (long long)nb);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
}  // namespace

extern "C" int catseg_lovasz_softmax(const float* logits, const int64_t* labels, long long P, int K, float weight,
                                     float* loss_out, float* dlogits, int accumulate_dlogits, void* workspace,
                                     size_t workspace_bytes, catseg_stream_t stream) {
  return lovasz_run(logits, labels, P, K, weight, loss_out, dlogits, dlogits != nullptr, accumulate_dlogits, workspace, workspace_bytes, stream);
}

// The same pipeline in two calls around autograd: _fwd leaves d loss / d prob (class planes) and the class counts in `workspace` when
// want_grad != 0; _bwd turns them into d loss / d logit times `upstream` (a DEVICE scalar: autograd's grad_output of the loss; null = 1) -- the
// pass over 4 P K bytes that used to multiply the stored gradient by the upstream scalar disappears.  The caller keeps workspace and
// logits untouched between the two calls.
extern "C" int catseg_lovasz_softmax_fwd(const float* logits, const int64_t* labels, long long P, int K, float weight, float* loss_out,
                                         int want_grad, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  return lovasz_run(logits, labels, P, K, weight, loss_out, nullptr, want_grad, 0, workspace, workspace_bytes, stream);
}

extern "C" int catseg_lovasz_softmax_bwd(const float* logits, long long P, int K, float weight, const float* upstream, float* dlogits,
                                         int accumulate_dlogits, void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(P > 0 && P < (1ll << 31) && K > 0 && K <= MAXK && logits && dlogits && workspace, "lovasz bwd: bad args");
  CS_REQUIRE(workspace_bytes >= lv_layout(P, K, nullptr, nullptr), "lovasz bwd: workspace too small");
  LvWs w;
  lv_layout(P, K, (char*)workspace, &w);
  const int nb = (int)((P + PIX - 1) / PIX);
  const size_t shb = (size_t)PIX * (K | 1) * 4;
  // (the workspace holds what THIS process' _fwd call left: catseg_debug_set_lovasz_prune must not change between the two calls)
  hipLaunchKernelGGL(lv_backward_kernel, dim3(nb), dim3(PIX), 2 * shb, (hipStream_t)stream, logits, P, K, (const uint32_t*)w.counts, (const float*)w.dprob, weight,
                     dlogits, accumulate_dlogits, upstream, g_prune ? (const unsigned long long*)w.actmask : nullptr, (const uint32_t*)w.blkcnt, (long long)nb);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" size_t catseg_ce_workspace(long long P) { return cs_align_up((size_t)((P + PIX - 1) / PIX) * 8 + 256, 256); }

extern "C" int catseg_cross_entropy(const float* logits, const int64_t* labels, long long P, int K, long long ignore_index,
                                    float weight, float* loss_out, float* dlogits, void* workspace,
                                    size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(P > 0 && K > 0 && K <= MAXK, "ce: bad args");
  if (workspace_bytes < catseg_ce_workspace(P) || !workspace) {
    catseg_set_error("ce: workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((P + PIX - 1) / PIX);
  const size_t shb = (size_t)PIX * (K | 1) * 4;
  float* inv = (float*)workspace;
  float* part = inv + 64;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(nb), dim3(PIX), shb, st, logits, labels, P, K, ignore_index, part);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, (long long)nb, weight, loss_out, inv);
  if (dlogits) hipLaunchKernelGGL(ce_bwd_kernel, dim3(nb), dim3(PIX), shb, st, logits, labels, P, K, ignore_index, weight, (const float*)inv, dlogits);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_confusion_matrix(const float* logits, const int64_t* labels, long long P, int K, int32_t* cm,
                                       catseg_stream_t stream) {
  CS_REQUIRE(P > 0 && K > 0 && K <= MAXK, "confusion: bad args");
  const int nb = (int)((P + PIX - 1) / PIX);
  const size_t shb = (size_t)PIX * (K | 1) * 4 + (size_t)K * K * 4;
  hipLaunchKernelGGL(confusion_kernel, dim3(nb), dim3(PIX), shb, (hipStream_t)stream, logits, labels, P, K, cm);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                                float beta2, float eps, int step, float grad_scale, catseg_stream_t stream) {
  CS_REQUIRE(n > 0 && step >= 1 && cs_aligned16(p) && cs_aligned16(g) && cs_aligned16(m) && cs_aligned16(v), "adam: bad args");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const long long n4 = (n + 3) / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, lr, beta1, beta2, eps, (float)bc1,
                     (float)sqrt(bc2), grad_scale, (const float*)nullptr);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" void catseg_adam_hyper(float lr, float beta1, float beta2, int step, float grad_scale, float* hyper4) {
  hyper4[0] = lr;
  hyper4[1] = (float)(1.0 - pow((double)beta1, step));
  hyper4[2] = (float)sqrt(1.0 - pow((double)beta2, step));
  hyper4[3] = grad_scale;
}

extern "C" int catseg_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1,
                                    float beta2, float eps, catseg_stream_t stream) {
  CS_REQUIRE(n > 0 && hyper && cs_aligned16(p) && cs_aligned16(g) && cs_aligned16(m) && cs_aligned16(v), "adam: bad args");
  const long long n4 = (n + 3) / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, 0.f, beta1, beta2, eps, 1.f, 1.f, 1.f,
                     hyper);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
