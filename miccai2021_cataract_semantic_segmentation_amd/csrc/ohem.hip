// OhemCrossEntropy (reference: losses/OhemCrossEntropy.py:22-39) on gfx950.
//
//   p_i      = softmax(score_i)[target_i]                      for every non-ignored pixel
//   min_val  = k-th smallest p, k = min(min_kept, n_valid - 1)  (the reference sorts all p; here: a 3-pass radix SELECT
//              on the fp32 bit pattern, 11 + 11 + 10 bits, histograms in LDS -> global, no sort, no host sync)
//   thr      = max(min_val, thresh)
//   loss     = mean over { i valid, p_i < thr } of CE_i ;  d loss / d score = (softmax - onehot) / count on that set
//
// All of it is HBM-bound: the logits are read twice (prep, backward), keys three more times (4 B / pixel / pass).
#include "common.h"

namespace {

constexpr int PIX = 256;
constexpr int MAXK = 64;
constexpr int NBIN = 2048;

struct OhemWs {
  uint32_t* keys;    // [P]   bit pattern of p (monotone for p >= 0), 0xFFFFFFFF for ignored pixels
  float* loss;       // [P]   per-pixel cross entropy
  uint32_t* hist;    // [NBIN]
  uint32_t* state;   // [0] prefix bits, [1] remaining rank, [2] n_valid, [3] threshold bits
  float* part;       // [2 * nblocks] partial (sum, count)
  float* inv;        // [1] 1 / count
};

size_t carve(long long P, void* base, OhemWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void* p = base ? (char*)base + off : nullptr;
    off += cs_align_up(bytes, 256);
    return p;
  };
  const long long nb = (P + PIX - 1) / PIX;
  OhemWs t;
  t.keys = (uint32_t*)take((size_t)P * 4);
  t.loss = (float*)take((size_t)P * 4);
  t.hist = (uint32_t*)take(NBIN * 4);
  t.state = (uint32_t*)take(64);
  t.part = (float*)take((size_t)nb * 8);
  t.inv = (float*)take(64);
  if (w) *w = t;
  return off;
}

__device__ __forceinline__ void stage_rows(const float* __restrict__ logits, long long p0, int np, int K, int KS, float* sh) {
  const int n = np * K;
  const float* src = logits + p0 * K;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / K, c = i - r * K;
    sh[r * KS + c] = src[i];
  }
}

// pass 0: per-pixel p / CE, keys + losses to HBM, histogram of the top 11 key bits
__global__ __launch_bounds__(PIX) void ohem_prep_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long long P, int K,
                                                        long long ignore, uint32_t* __restrict__ keys, float* __restrict__ loss,
                                                        uint32_t* __restrict__ hist) {
  extern __shared__ float sh[];
  __shared__ uint32_t h[NBIN];
  const int KS = K | 1;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  for (int i = threadIdx.x; i < NBIN; i += PIX) h[i] = 0;
  stage_rows(logits, p0, np, K, KS, sh);
  __syncthreads();
  const int t = threadIdx.x;
  if (t < np) {
    const int64_t lab = labels[p0 + t];
    uint32_t key = 0xFFFFFFFFu;
    float l = 0.f;
    if (lab != ignore && lab >= 0 && lab < K) {
      const float* row = sh + t * KS;
      float m = row[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
      float s = 0.f;
      for (int c = 0; c < K; ++c) s += expf(row[c] - m);
      const float z = row[(int)lab] - m;
      l = logf(s) - z;
      key = __float_as_uint(expf(z) / s);
      atomicAdd(&h[key >> 21], 1u);
    }
    keys[p0 + t] = key;
    loss[p0 + t] = l;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NBIN; i += PIX)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

// one block: locate the bin that holds the wanted rank, extend the prefix, clear the histogram for the next pass
__global__ __launch_bounds__(256) void ohem_select_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ state, int pass, long long min_kept,
                                                          float thresh) {
  __shared__ uint32_t part[256];
  __shared__ uint32_t base_s;
  const int t = threadIdx.x;
  const int nbin = pass == 2 ? 1024 : NBIN;
  const int per = nbin / 256;
  uint32_t loc[8];
  uint32_t s = 0;
  for (int i = 0; i < per; ++i) { loc[i] = hist[t * per + i]; s += loc[i]; }
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; }
    base_s = run;  // total
  }
  __syncthreads();
  const uint32_t total = base_s;
  uint32_t rank;
  if (pass == 0) {
    const long long k = min_kept < (long long)total - 1 ? min_kept : (long long)total - 1;
    rank = (uint32_t)(k < 0 ? 0 : k);
  } else {
    rank = state[1];
  }
  __syncthreads();
  uint32_t run = part[t];
  for (int i = 0; i < per; ++i) {
    if (rank >= run && rank < run + loc[i]) {  // exactly one (thread, i) matches when total > 0
      const uint32_t bin = (uint32_t)(t * per + i);
      const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
      const uint32_t prefix = (pass == 0 ? 0u : state[0]) | (bin << shift);
      state[0] = prefix;
      state[1] = rank - run;
      if (pass == 2) state[3] = __float_as_uint(fmaxf(__uint_as_float(prefix), thresh));
    }
    run += loc[i];
  }
  if (pass == 0 && t == 0) {
    state[2] = total;
    if (total == 0) state[3] = __float_as_uint(thresh);  // the reference raises here (index -1 of an empty tensor); the mean below is nan
  }
  for (int i = 0; i < per; ++i) hist[t * per + i] = 0;
}

// passes 1 / 2: histogram of the next digit of the keys that match the prefix found so far
__global__ __launch_bounds__(256) void ohem_hist_kernel(const uint32_t* __restrict__ keys, long long P, const uint32_t* __restrict__ state, int pass,
                                                        uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[NBIN];
  for (int i = threadIdx.x; i < NBIN; i += 256) h[i] = 0;
  __syncthreads();
  const uint32_t prefix = state[0];
  const int hi_shift = pass == 1 ? 21 : 10;
  const int shift = pass == 1 ? 10 : 0;
  const uint32_t mask = pass == 1 ? 0x7FFu : 0x3FFu;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const uint32_t k = keys[i];
    if (k != 0xFFFFFFFFu && (k >> hi_shift) == (prefix >> hi_shift)) atomicAdd(&h[(k >> shift) & mask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NBIN; i += 256)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

__global__ __launch_bounds__(PIX) void ohem_sum_kernel(const uint32_t* __restrict__ keys, const float* __restrict__ loss, long long P,
                                                       const uint32_t* __restrict__ state, float* __restrict__ part) {
  const long long i = (long long)blockIdx.x * PIX + threadIdx.x;
  const uint32_t thr = state[3];
  float l = 0.f, c = 0.f;
  if (i < P) {
    const uint32_t k = keys[i];
    if (k != 0xFFFFFFFFu && k < thr) { l = loss[i]; c = 1.f; }  // p >= 0: unsigned order of the bits = order of the floats
  }
  l = wave_sum(l);
  c = wave_sum(c);
  __shared__ float r[8];
  const int t = threadIdx.x;
  if ((t & 63) == 0) { r[t >> 6] = l; r[4 + (t >> 6)] = c; }
  __syncthreads();
  if (t == 0) {
    part[2 * blockIdx.x] = r[0] + r[1] + r[2] + r[3];
    part[2 * blockIdx.x + 1] = r[4] + r[5] + r[6] + r[7];
  }
}

__global__ __launch_bounds__(256) void ohem_finalize_kernel(const float* __restrict__ part, long long nb, float weight, float* __restrict__ loss_out,
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
    loss_out[0] = (float)(s1[0] / s2[0]) * weight;  // mean of an empty selection = nan, as torch
    inv_count[0] = s2[0] > 0 ? (float)(1.0 / s2[0]) : 0.f;
  }
}

__global__ __launch_bounds__(PIX) void ohem_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long long P, int K,
                                                       const uint32_t* __restrict__ keys, const uint32_t* __restrict__ state, float weight,
                                                       const float* __restrict__ inv_count, float* __restrict__ dlogits) {
  extern __shared__ float sh[];
  const int KS = K | 1;
  const long long p0 = (long long)blockIdx.x * PIX;
  const int np = (int)min((long long)PIX, P - p0);
  stage_rows(logits, p0, np, K, KS, sh);
  __syncthreads();
  const int t = threadIdx.x;
  if (t < np) {
    float* row = sh + t * KS;
    const uint32_t k = keys[p0 + t];
    if (k != 0xFFFFFFFFu && k < state[3]) {
      const int lab = (int)labels[p0 + t];
      float m = row[0];
      for (int c = 1; c < K; ++c) m = fmaxf(m, row[c]);
      float s = 0.f;
      for (int c = 0; c < K; ++c) { const float e = expf(row[c] - m); row[c] = e; s += e; }
      const float w = weight * inv_count[0];
      for (int c = 0; c < K; ++c) row[c] = (row[c] / s - (c == lab ? 1.f : 0.f)) * w;
    } else {
      for (int c = 0; c < K; ++c) row[c] = 0.f;
    }
  }
  __syncthreads();
  const int n = np * K;
  float* dst = dlogits + p0 * K;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int r = i / K, c = i - r * K;
    dst[i] = sh[r * KS + c];
  }
}

}  // namespace

extern "C" size_t catseg_ohem_workspace(long long P) { return carve(P, nullptr, nullptr); }

extern "C" int catseg_ohem_cross_entropy(const float* logits, const int64_t* labels, long long P, int K, long long ignore_index,
                                         float thresh, long long min_kept, float weight, float* loss_out, float* dlogits,
                                         void* workspace, size_t workspace_bytes, catseg_stream_t stream) {
  CS_REQUIRE(P > 0 && P < (1ll << 32) - 1 && K > 0 && K <= MAXK, "ohem: need 0 < P < 2^32 - 1 and K <= %d", MAXK);
  CS_REQUIRE(min_kept >= 0 && thresh >= 0.f, "ohem: min_kept and thresh must be >= 0");
  if (workspace_bytes < carve(P, nullptr, nullptr) || !workspace) {
    catseg_set_error("ohem: workspace too small");
    return CATSEG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  OhemWs w;
  carve(P, workspace, &w);
  const long long nb = (P + PIX - 1) / PIX;
  const size_t shb = (size_t)PIX * (K | 1) * 4;
  if (hipMemsetAsync(w.hist, 0, NBIN * 4 + 64, st) != hipSuccess) { catseg_set_error("ohem: memset failed"); return CATSEG_EHIP; }
  hipLaunchKernelGGL(ohem_prep_kernel, dim3(nb), dim3(PIX), shb, st, logits, labels, P, K, ignore_index, w.keys, w.loss, w.hist);
  const int hb = (int)(nb < 2048 ? nb : 2048);
  for (int pass = 0; pass < 3; ++pass) {
    if (pass > 0) hipLaunchKernelGGL(ohem_hist_kernel, dim3(hb), dim3(256), 0, st, (const uint32_t*)w.keys, P, (const uint32_t*)w.state, pass, w.hist);
    hipLaunchKernelGGL(ohem_select_kernel, dim3(1), dim3(256), 0, st, w.hist, w.state, pass, min_kept, thresh);
  }
  hipLaunchKernelGGL(ohem_sum_kernel, dim3(nb), dim3(PIX), 0, st, (const uint32_t*)w.keys, (const float*)w.loss, P, (const uint32_t*)w.state, w.part);
  hipLaunchKernelGGL(ohem_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)w.part, nb, weight, loss_out, w.inv);
  if (dlogits)
    hipLaunchKernelGGL(ohem_bwd_kernel, dim3(nb), dim3(PIX), shb, st, logits, labels, P, K, (const uint32_t*)w.keys, (const uint32_t*)w.state, weight,
                       (const float*)w.inv, dlogits);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
