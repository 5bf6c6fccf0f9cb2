// DIRECT 3x3 / stride 1 / pad 1 convolution in split precision for the HRNet trunk (Cin = Cout = 48, 96: BasicBlock convolutions,
// reference models/HRNetv2.py:22-65; 128 launches forward + 128 backward-data per OCRNet-HRNet-W48 step).
//
// As an implicit GEMM these layers re-read every input pixel nine times through L2 -> LDS (once per filter tap) and spend a
// separate pass on splitting the activation into bf16 planes.  Here a block owns a TH x TW pixel tile of one image:
//   * the fp32 halo tile (TH+2) x (TW+2) x KC channels is read ONCE (global -> VGPR, prefetched one channel chunk ahead), split
//     exactly into three bf16 planes IN the kernel (x = h + m + l) and written to LDS;
//   * all nine taps are shifted windows of that LDS image (the tap shift is an immediate offset of the fragment read);
//   * the weights are a pre-arranged, pre-split image [co block][chunk][K-step][plane][k-group][co][8] that streams through a
//     double-buffered LDS slot by LDS-DMA, one 32-deep K-step ahead of the MFMAs;
//   * v_mfma_f32_16x16x32_bf16, D[co][px]: the weights are the A operand, the pixels the B operand, so that a lane ends up with
//     four consecutive output channels of one pixel (16-byte stores into the NHWC result); six products per (co, px, K-step)
//     block (hh hl lh hm mh mm) accumulate in fp32 -- the arithmetic of igemm_bf16x3.hip;
//   * epilogue: bias, BatchNorm partial statistics per (tile, channel) for catseg_bn_finalize_counts, optional accumulate.
// Backward-data of the same layer is the same kernel on dy with the transposed, tap-mirrored weight image.
//
// LDS images (conflict-free ds_read_b128 by construction): 8-channel groups are the OUTER index, [k-group][halo pixel][16 B],
// the k-group stride a multiple of 256 B: the 16 lanes that ds_read_b128 serves per cycle hold 16 different pixels (or output
// channels) of at most two k-groups and therefore 16 different 16-byte slots of the 256-byte bank row.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// K-step tables.  A K-step is 32 deep = two units of 16 channels: unit = (tap, 16-channel window of the chunk).  Chunks of 32
// channels: step s = tap s, windows 0 / 1.  Chunks of 48 channels: steps 0-8 as before, steps 9-12 pair taps (2j, 2j+1) of window
// 2, step 13 = (tap 8, window 2) + an empty unit (zero weights).
struct Unit { int tap, win, live; };
__host__ __device__ constexpr Unit unit_of(int KC, int step, int half) {
  if (step < 9) return Unit{step, half, 1};
  (void)KC;
  const int t = 2 * (step - 9) + half;
  return t < 9 ? Unit{t, 2, 1} : Unit{8, 2, 0};
}
__host__ __device__ constexpr int steps_of(int KC) { return KC == 32 ? 9 : 14; }

template <int C_, int KC_, int NT_, int WC_, int WP_, int PB_, int TPH_, int TPW_, int PH_, int PW_>
struct DcCfg {
  static constexpr int C = C_, KC = KC_, NT = NT_, WC = WC_, WP = WP_, PB = PB_, TPH = TPH_, TPW = TPW_, PH = PH_, PW = PW_;
  static_assert(KC == 32 || KC == 48, "channel chunk");
  static_assert(C % KC == 0 && C % NT == 0 && NT % (16 * WC) == 0 && PH * PW == 16 && TPH * TPW == WP * PB, "tiling");
  static constexpr int NW = WC * WP, NTHR = 64 * NW;
  static constexpr int CB = NT / 16 / WC;                 // output-channel tiles per wave
  static constexpr int TH = TPH * PH, TW = TPW * PW;       // pixel tile of a block
  static constexpr int HH = TH + 2, HW = TW + 2;           // halo tile
  // halo row stride in pixels (= 16-byte slots): the PH rows of a pixel tile must land on disjoint slot groups
  static constexpr int RW = PH == 1 ? HW : ((HW - PW + 15) / 16 * 16 + PW);
  static constexpr int HP = (HH * RW + 15) / 16 * 16;
  static constexpr int NKG = KC / 8;
  static constexpr int KGS = HP * 16;                      // bytes between 8-channel groups (multiple of 256)
  static constexpr int XPS = NKG * KGS;                    // bytes between planes
  static constexpr int XBYTES = 3 * XPS;
  static constexpr int WPS = 64 * NT;                      // bytes per plane of one K-step of weights: [4 k-groups][NT][16 B]
  static constexpr int WSTEP = 3 * WPS;
  static constexpr int NCHUNK = C / KC, NSTEP = steps_of(KC);
  static constexpr int NITEM = HH * HW * NKG;              // staging items: (halo pixel, 8-channel group)
  static constexpr int IPT = (NITEM + NTHR - 1) / NTHR;
  static constexpr int WITEMS = WSTEP / 1024;              // LDS-DMA wave instructions per K-step
  static_assert(WSTEP % 1024 == 0, "weights of a K-step in whole 1 KB pieces");
  static constexpr int LDS = XBYTES + 2 * WSTEP;
  static constexpr int SCR = 3 * WP * NT * 4;              // epilogue scratch (floats -> bytes), inside the X image
  static_assert(SCR <= XBYTES, "epilogue scratch");
};

struct DcArgs {
  const float* x;
  int ldx;
  const u16* wimg;
  float* y;
  int ldy;
  const float* bias;
  int B, H, W;
  int tiles_y, tiles_x;
  int accumulate;
  float* bn_part;   // [tile][3][C] or nullptr
  int* bn_cnt;      // [tile] valid pixels
};

__device__ __forceinline__ void dc_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <class G>
__global__ __launch_bounds__(G::NTHR, 2) void dconv3_b3_kernel(const DcArgs a) {
  __shared__ __attribute__((aligned(256))) unsigned char smem[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / G::WP, wp = wave % G::WP;
  const int i16 = lane & 15, kg = lane >> 4;

  // ---- tile of this block; each XCD (blockIdx.x % 8 labels the blocks that share one) gets a contiguous run of tiles ----------
  const int ntile = gridDim.x;
  int tile;
  {
    const int q = ntile >> 3, r = ntile & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int tx = tile % a.tiles_x;
  const int ty = (tile / a.tiles_x) % a.tiles_y;
  const int b = tile / (a.tiles_x * a.tiles_y);
  const int y0 = ty * G::TH, x0 = tx * G::TW;
  const int cob = blockIdx.y;                       // output-channel block
  const long long img0 = (long long)b * a.H * a.W;  // first pixel of the image

  // ---- staging items: clamped source offset (always in bounds), validity, LDS destination -----------------------------------
  int s_off[G::IPT], s_dst[G::IPT];
  bool s_ok[G::IPT];
#pragma unroll
  for (int i = 0; i < G::IPT; ++i) {
    const int q = tid + i * G::NTHR;
    const int hp = q / G::NKG, g8 = q % G::NKG;
    const int hy = hp / G::HW, hx = hp % G::HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    s_ok[i] = q < G::NITEM && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
    s_off[i] = (cy * a.W + cx) * a.ldx + g8 * 8;
    s_dst[i] = q < G::NITEM ? g8 * G::KGS + (hy * G::RW + hx) * 16 : -1;
  }
  const float* xin = a.x + img0 * a.ldx;
  f32x4 pre[G::IPT][2];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < G::IPT; ++i) {
      const float* p = xin + s_off[i] + chunk * G::KC;
      pre[i][0] = *(const f32x4*)p;
      pre[i][1] = *(const f32x4*)(p + 4);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < G::IPT; ++i) {
      if (s_dst[i] < 0) continue;
      bf16x8 h, m, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = s_ok[i] ? pre[i][j >> 2][j & 3] : 0.f;
        const __bf16 hh = (__bf16)v;
        const float r1 = v - (float)hh;
        const __bf16 mm = (__bf16)r1;
        h[j] = hh;
        m[j] = mm;
        l[j] = (__bf16)(r1 - (float)mm);
      }
      *(bf16x8*)(smem + s_dst[i]) = h;
      *(bf16x8*)(smem + s_dst[i] + G::XPS) = m;
      *(bf16x8*)(smem + s_dst[i] + 2 * G::XPS) = l;
    }
  };

  // ---- weight stream: K-step t (over all chunks) of this co block -> LDS slot t & 1 ------------------------------------------
  const unsigned char* wsrc = (const unsigned char*)a.wimg + (long long)cob * (G::NCHUNK * G::NSTEP) * G::WSTEP + lane * 16;
  auto wfill = [&](int t) {
    unsigned char* dst = smem + G::XBYTES + (t & 1) * G::WSTEP;
    const unsigned char* src = wsrc + (long long)t * G::WSTEP;
#pragma unroll
    for (int i = 0; i < (G::WITEMS + G::NW - 1) / G::NW; ++i) {
      const int it = wave + i * G::NW;
      if (it < G::WITEMS) dc_glds16(src + it * 1024, dst + it * 1024);
    }
  };

  // ---- fragment addresses --------------------------------------------------------------------------------------------------
  // pixel tile pt of this wave: lane i16 -> pixel (pr, pc) of the tile; halo coordinates of tap (ky, kx) = (row + ky, col + kx)
  int xb[G::PB], xb2[G::PB];
  bool p_ok[G::PB];
  int p_row[G::PB], p_col[G::PB];
#pragma unroll
  for (int pt = 0; pt < G::PB; ++pt) {
    const int pl = wp * G::PB + pt;
    const int row = (pl / G::TPW) * G::PH + i16 / G::PW, col = (pl % G::TPW) * G::PW + i16 % G::PW;
    p_row[pt] = row;
    p_col[pt] = col;
    p_ok[pt] = y0 + row < a.H && x0 + col < a.W;
    xb[pt] = kg * G::KGS + (row * G::RW + col) * 16;
    xb2[pt] = (kg & 1) * G::KGS + (row * G::RW + col) * 16;
  }
  const int wb = kg * (G::NT * 16) + (wc * G::CB * 16 + i16) * 16;

  f32x4 acc[G::CB][G::PB];
#pragma unroll
  for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto kstep = [&](const int s, const unsigned char* wbuf) {     // (s is a compile-time constant after unrolling)
    const Unit ua = unit_of(G::KC, s, 0), ub = unit_of(G::KC, s, 1);
    const int offa = ((ua.tap / 3) * G::RW + ua.tap % 3) * 16, offb = ((ub.tap / 3) * G::RW + ub.tap % 3) * 16;
    bf16x8 wf[G::CB][3], xf[G::PB][3];
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
      for (int p = 0; p < 3; ++p) wf[ct][p] = *(const bf16x8*)(wbuf + p * G::WPS + wb + ct * 256);
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) {
      int o;
      if (s < 9) o = xb[pt] + offa + 2 * ua.win * G::KGS;
      else o = xb2[pt] + 2 * ua.win * G::KGS + (kg >> 1 ? offb : offa);
#pragma unroll
      for (int p = 0; p < 3; ++p) xf[pt][p] = *(const bf16x8*)(smem + p * G::XPS + o);
    }
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
      for (int pt = 0; pt < G::PB; ++pt) {
        f32x4 c = acc[ct][pt];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][1], xf[pt][1], c, 0, 0, 0);   // m m
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][0], xf[pt][2], c, 0, 0, 0);   // h l
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][2], xf[pt][0], c, 0, 0, 0);   // l h
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][0], xf[pt][1], c, 0, 0, 0);   // h m
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][1], xf[pt][0], c, 0, 0, 0);   // m h
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][0], xf[pt][0], c, 0, 0, 0);   // h h
        acc[ct][pt] = c;
      }
  };

  // ---- main loop ------------------------------------------------------------------------------------------------------------
  fetch(0);
  wfill(0);
  int t = 0;
#pragma unroll 1
  for (int chunk = 0; chunk < G::NCHUNK; ++chunk) {
    stash();                                   // (the previous chunk's last reads are behind the barrier that closed its last K-step)
    if (chunk + 1 < G::NCHUNK) fetch(chunk + 1);
    __syncthreads();                           // chunk image + weights of K-step t visible (the barrier drains the LDS-DMA)
#pragma unroll
    for (int s = 0; s < G::NSTEP; ++s) {
      const bool more = !(s == G::NSTEP - 1 && chunk == G::NCHUNK - 1);
      if (more) wfill(t + 1);
      kstep(s, smem + G::XBYTES + (t & 1) * G::WSTEP);
      ++t;
      __syncthreads();
    }
  }

  // ---- epilogue: bias, store, BatchNorm partials ------------------------------------------------------------------------------
  const int co0 = cob * G::NT + wc * G::CB * 16 + 4 * kg;   // + ct * 16 + r
#pragma unroll
  for (int ct = 0; ct < G::CB; ++ct) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *(const f32x4*)(a.bias + co0 + ct * 16);
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) {
      acc[ct][pt] += bv;
      if (p_ok[pt]) {
        float* dst = a.y + (img0 + (long long)(y0 + p_row[pt]) * a.W + x0 + p_col[pt]) * a.ldy + co0 + ct * 16;
        f32x4 v = acc[ct][pt];
        if (a.accumulate) v += *(const f32x4*)dst;
        *(f32x4*)dst = v;
      }
    }
  }
  if (a.bn_part) {
    // (tile mean, sum(v - mean), sum((v - mean)^2)) per channel over the tile's valid pixels: two in-register passes; the 16 lanes
    // of a k-group hold 16 pixels of the same four channels, the WP waves of a channel group the other pixels
    float* scr = (float*)smem;    // [3][WP][NT]  (the K loop's last barrier is behind every LDS read)
    const int nvalid = min(G::TH, a.H - y0) * min(G::TW, a.W - x0);
    const float inv = 1.f / (float)nvalid;
    const int cl = wc * G::CB * 16 + 4 * kg;   // block-local channel of (ct = 0, r = 0)
    float s0[G::CB][4];
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.f;
#pragma unroll
        for (int pt = 0; pt < G::PB; ++pt) v += p_ok[pt] ? acc[ct][pt][r] : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        s0[ct][r] = v;
      }
    if (i16 == 0) {
#pragma unroll
      for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) scr[wp * G::NT + cl + ct * 16 + r] = s0[ct][r];
    }
    __syncthreads();
    float mean[G::CB][4];
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < G::WP; ++w) v += scr[w * G::NT + cl + ct * 16 + r];
        mean[ct][r] = v * inv;
      }
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int pt = 0; pt < G::PB; ++pt) {
          const float d = p_ok[pt] ? acc[ct][pt][r] - mean[ct][r] : 0.f;
          d1 += d;
          d2 += d * d;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          d1 += __shfl_xor(d1, o, 64);
          d2 += __shfl_xor(d2, o, 64);
        }
        if (i16 == 0) {
          scr[wp * G::NT + cl + ct * 16 + r] = d1;
          scr[(G::WP + wp) * G::NT + cl + ct * 16 + r] = d2;
        }
      }
    __syncthreads();
    if (wp == 0 && i16 == 0) {
      float* part = a.bn_part + (long long)tile * 3 * G::C + cob * G::NT;
#pragma unroll
      for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float d1 = 0.f, d2 = 0.f;
#pragma unroll
          for (int w = 0; w < G::WP; ++w) {
            d1 += scr[w * G::NT + cl + ct * 16 + r];
            d2 += scr[(G::WP + w) * G::NT + cl + ct * 16 + r];
          }
          const int c = cl + ct * 16 + r;
          part[c] = mean[ct][r];
          part[G::C + c] = d1;
          part[2 * G::C + c] = d2;
        }
    }
    if (tid == 0 && cob == 0) a.bn_cnt[tile] = nvalid;
  }
}

// OHWI fp32 weights [C][3][3][C] -> the kernel's weight image, split into three bf16 planes.
//   forward:       A[co][k = (tap, c)]  = w[co][tap][c]
//   backward-data: A[ci][k = (tap, o)]  = w[o][8 - tap][ci]      (dx = conv of dy with the transposed, tap-mirrored bank)
__global__ __launch_bounds__(256) void dconv3_prep_kernel(const float* __restrict__ w, int C, int KC, int NT, int dgrad, u16* __restrict__ img) {
  const int nstep = steps_of(KC), nchunk = C / KC;
  const long long total = (long long)(C / NT) * nchunk * nstep * NT * 32;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7);
    long long r = i >> 3;
    const int co = (int)(r % NT); r /= NT;
    const int g = (int)(r & 3); r >>= 2;
    const int step = (int)(r % nstep); r /= nstep;
    const int chunk = (int)(r % nchunk);
    const int cob = (int)(r / nchunk);
    const Unit u = unit_of(KC, step, g >> 1);
    const int c = chunk * KC + u.win * 16 + (g & 1) * 8 + j;
    const int o = cob * NT + co;
    float v = 0.f;
    if (u.live) v = dgrad ? w[((long long)c * 9 + (8 - u.tap)) * C + o] : w[((long long)o * 9 + u.tap) * C + c];
    const __bf16 hh = (__bf16)v;
    const float r1 = v - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const __bf16 ll = (__bf16)(r1 - (float)mm);
    const long long base = (((long long)(cob * nchunk + chunk) * nstep + step) * 3) * (32LL * NT) + ((long long)g * NT + co) * 8 + j;
    img[base] = __builtin_bit_cast(u16, hh);
    img[base + 32LL * NT] = __builtin_bit_cast(u16, mm);
    img[base + 64LL * NT] = __builtin_bit_cast(u16, ll);
  }
}

//                     C  KC  NT WC WP PB TPH TPW PH PW
using Cfg48 = DcCfg<48, 48, 48, 1, 4, 2, 8, 1, 1, 16>;    // tile  8 x 16, wave = 48 co x 32 px
using Cfg96 = DcCfg<96, 32, 96, 2, 2, 4, 4, 2, 1, 16>;    // tile  4 x 32, wave = 48 co x 64 px

struct DcPlan { int kind, KC, NT, TH, TW; };

DcPlan dc_plan(int C) {
  if (C == 48) return {1, Cfg48::KC, Cfg48::NT, Cfg48::TH, Cfg48::TW};
  if (C == 96) return {2, Cfg96::KC, Cfg96::NT, Cfg96::TH, Cfg96::TW};
  return {0, 0, 0, 0, 0};
}

template <class G>
int dc_launch(const DcArgs& a, int C, hipStream_t st) {
  static bool attr_done = false;
  (void)attr_done;
  const int ntile = a.B * a.tiles_y * a.tiles_x;
  hipLaunchKernelGGL((dconv3_b3_kernel<G>), dim3(ntile, C / G::NT), dim3(G::NTHR), 0, st, a);
  return 0;
}

}  // namespace

extern "C" int catseg_dconv3_supported(int C) { return dc_plan(C).kind != 0; }

extern "C" size_t catseg_dconv3_wimg_bytes(int C) {
  const DcPlan p = dc_plan(C);
  if (!p.kind) return 0;
  return (size_t)(C / p.NT) * (C / p.KC) * steps_of(p.KC) * 192 * p.NT;
}

extern "C" int catseg_dconv3_tiles(int C, int B, int H, int W, int* tile_h, int* tile_w) {
  const DcPlan p = dc_plan(C);
  if (!p.kind) return 0;
  if (tile_h) *tile_h = p.TH;
  if (tile_w) *tile_w = p.TW;
  return B * ((H + p.TH - 1) / p.TH) * ((W + p.TW - 1) / p.TW);
}

extern "C" int catseg_dconv3_prep(const float* w, int C, int backward_data, void* wimg, catseg_stream_t stream) {
  const DcPlan p = dc_plan(C);
  CS_REQUIRE(p.kind && w && wimg, "dconv3 prep: unsupported channel count %d", C);
  const long long total = (long long)(C / p.NT) * (C / p.KC) * steps_of(p.KC) * p.NT * 32;
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(dconv3_prep_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, (hipStream_t)stream, w, C, p.KC, p.NT,
                     backward_data ? 1 : 0, (u16*)wimg);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_dconv3(int B, int H, int W, int C, const float* x, int ldx, const void* wimg, const float* bias, float* y, int ldy,
                             int accumulate, float* bn_part, size_t bn_part_floats, int* bn_counts, catseg_stream_t stream) {
  const DcPlan p = dc_plan(C);
  CS_REQUIRE(p.kind, "dconv3: unsupported channel count %d", C);
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && x && wimg && y, "dconv3: bad args");
  CS_REQUIRE(ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && cs_aligned16(x) && cs_aligned16(y) && cs_aligned16(wimg) &&
                 cs_aligned16(bias), "dconv3: alignment / row strides");
  CS_REQUIRE((long long)H * W * (long long)(ldx > ldy ? ldx : ldy) < (1LL << 31), "dconv3: image too large for 32-bit offsets");
  DcArgs a;
  a.x = x; a.ldx = ldx; a.wimg = (const u16*)wimg; a.y = y; a.ldy = ldy; a.bias = bias;
  a.B = B; a.H = H; a.W = W;
  a.tiles_y = (H + p.TH - 1) / p.TH;
  a.tiles_x = (W + p.TW - 1) / p.TW;
  a.accumulate = accumulate;
  a.bn_part = bn_part;
  a.bn_cnt = bn_counts;
  const long long ntile = (long long)B * a.tiles_y * a.tiles_x;
  if (bn_part) CS_REQUIRE(bn_counts && bn_part_floats >= (size_t)ntile * 3 * C, "dconv3: BatchNorm partial buffer too small");
  if (p.kind == 1) dc_launch<Cfg48>(a, C, (hipStream_t)stream);
  else dc_launch<Cfg96>(a, C, (hipStream_t)stream);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
