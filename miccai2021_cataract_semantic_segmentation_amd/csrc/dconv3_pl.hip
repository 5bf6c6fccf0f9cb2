// DIRECT 3x3 / stride 1 / pad 1 convolution of the HRNet trunk widths on PRODUCER-WRITTEN fp16 x 2 operand planes (round 4).
// Reference layers: conv3x3(planes, planes) of BasicBlock / Bottleneck, models/HRNetv2.py:22-65 (208 forward + 208 backward-data launches per
// OCRNet-HRNet-W48 step).
//
// dconv3_f16x2.hip reads the fp32 activation, scales and splits it into two fp16 planes in registers and stores the planes to LDS -- in
// the same waves that issue the MFMAs: 4.5 VALU instructions per MFMA, matrix pipes busy 32 % (profiles/r03_pmc_sq_dconv3_h2_48.json), and a
// two-slot weight stream whose LDS-DMA is issued and awaited inside ONE K-step (every step >= one L2 round trip: ~1000 cycles against
// 288 cycles of MFMAs).  Here the planes already exist in HBM (csrc/planes.h: [plane][channel group][pixel][8], written by the tensor's
// producer with an exponent derived from a bound), so that
//   * the halo tile of a pixel tile streams into LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane = one (pixel, channel group)),
//     straight into the [group][halo pixel][16 B] image the fragment reads want: NO VALU work, no register staging, no ds_write;
//     padding = lanes whose pixel lies outside the image get an out-of-range offset, for which the hardware writes zeros;
//   * 4 HELPER waves per block issue every LDS-DMA (weights three K-steps ahead through three slots, the next channel chunk's halo image
//     one chunk ahead through a two-buffer ring) and wait for them; 4 COMPUTE waves only read fragments and issue MFMAs.  Both roles meet
//     at ONE raw s_barrier per K-step; the chunks of consecutive tiles form one continuous stream (the epilogue of a tile runs while the
//     helpers prefetch the next tile);
//   * the epilogue needs no LDS and no barrier: BatchNorm partials are written per WAVE (a row of the partial buffer per (tile, pixel
//     group): catseg_bn_finalize_counts merges rows with individual pixel counts anyway), max|y| goes to the output's amax record.
// Arithmetic: exactly dconv3_f16x2's (v_mfma_f32_16x16x32_f16, products hl, lh, hh into one fp32 accumulator, result scaled by
// 2^-(e_x + e_w)); the weight images and their records are the ones catseg_dconv3_f16x2_prep_batch writes.
#include <type_traits>
#include "planes.h"

int catseg_g_pl_slots = 512;
int catseg_g_pl_pair = 0;          // bit mask (2: 96 channels, 4: 192, 8: 384): launch the two-tiles-per-block form of the kernel
extern "C" int catseg_debug_set_dconv3_pl_pair(int mask) {
  catseg_g_pl_pair = mask;
  return CATSEG_OK;
}
extern "C" int catseg_debug_set_dconv3_pl_slots(int slots) {
  catseg_g_pl_slots = slots > 0 ? slots : 512;
  return CATSEG_OK;
}

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

// K-step tables of the weight image (csrc/dconv3_b3.hip: unit_of): a K-step is 32 deep = two units (tap, 16-channel window)
struct PlUnit { int tap, win, live; };
constexpr PlUnit pl_unit_of(int step, int half) {
  if (step < 9) return PlUnit{step, half, 1};
  const int t = 2 * (step - 9) + half;
  return t < 9 ? PlUnit{t, 2, 1} : PlUnit{8, 2, 0};
}
constexpr int pl_steps_of(int KC) { return KC == 32 ? 9 : 14; }

// NT2_: pixel tiles a block works on SIDE BY SIDE (1: four compute waves, two blocks per CU; 2: eight compute waves on two tiles that SHARE the
// weight slots -- one stream of weights per CU instead of two -- and one block per CU)
template <int C_, int NT_, int WC_, int WP_, int PB_, int TPH_, int TPW_, int NT2_ = 1>
struct PlCfg {
  static constexpr int C = C_, NT = NT_, WC = WC_, WP = WP_, PB = PB_, TPH = TPH_, TPW = TPW_, NT2 = NT2_;
  static constexpr int NCW = 4 * NT2, NTHR = 64 * (NCW + 4), MINW = NT2 == 1 ? 4 : 3;
  static constexpr int KC = C == 48 ? 48 : 32;            // channel chunk of the WEIGHT image (as dconv3_f16x2 lays it out)
  static_assert(WC * WP == 4 && C % KC == 0 && C % NT == 0 && NT % (16 * WC) == 0 && TPH * TPW == WP * PB, "tiling");
  static constexpr int NG = C / 8;                        // channel groups of the planes
  static constexpr int CB = NT / 16 / WC;                 // output-channel tiles per wave
  static constexpr int TH = TPH, TW = TPW * 16, HH = TH + 2, HW = TW + 2;
  static constexpr int NPX = HH * HW, HP = (NPX + 15) / 16 * 16;
  static constexpr int KGS = HP * 16, XPS = 4 * KGS, XBUF = 2 * XPS;     // one ring buffer: [plane][4 groups][HP][16 B]
  static constexpr int NJ = (HP + 63) / 64;                               // LDS-DMA instructions per (plane, group)
  static constexpr int WPS = 64 * NT, WSTEP = 2 * WPS, WITEMS = WSTEP / 1024;
  static constexpr int NCHUNK = C / KC, NSTEP = pl_steps_of(KC), TSTEPS = NCHUNK * NSTEP;
  // X sub-chunks of a weight chunk: KC = 32: one (4 groups, 9 K-steps); KC = 48: (4 groups, steps 0-8) + (2 groups, steps 9-13)
  static constexpr int NSUB = KC == 48 ? 2 : 1;
  static constexpr int NXB = 2;
  static constexpr int RING = NT2 * XBUF;                   // one ring buffer holds the halo images of the block's NT2 tiles
  static constexpr int BUDGET = (NT2 == 1 ? 80 : 128) * 1024;
  // weight slots: as many (3 .. 5) as fit two blocks per CU (80 KB each).  With NWB slots the weights of K-step s + NWB are issued in step s
  // and must have landed NWB - 2 steps later: the helpers' counted waits then cover an L2 round trip under load (~1 - 2 K-steps)
#ifdef PL_NWB
  static constexpr int NWB = PL_NWB;
#else
  static constexpr int NWB = (NXB * RING + 5 * WSTEP <= BUDGET) ? 5 : ((NXB * RING + 4 * WSTEP <= BUDGET) ? 4 : 3);
#endif
  static constexpr int LDS = NXB * RING + NWB * WSTEP;
  static_assert(WSTEP % 1024 == 0 && KGS % 256 == 0, "LDS image strides");
  static constexpr int sub_steps(int u) { return u == 0 ? 9 : 5; }
  static constexpr int sub_groups(int u) { return u == 0 ? 4 : 2; }
  static constexpr int tile_items(int u) { return u == 0 ? 2 * NJ : NJ; }  // X LDS-DMA instructions per helper wave and tile
  static constexpr int sub_items(int u) { return NT2 * tile_items(u); }
};

struct PlArgs {
  const unsigned char* planes;   // [2][C / 8][P][8] fp16
  unsigned P16;                  // P * 16: bytes of one (plane, group) slab
  const int* x_rec;              // record of the planes: word CS_REC_EXP = the exponent they were written with
  const u16* wimg;
  const int* w_rec;              // record of the weight image: word 1 = exponent
  float* y;
  int ldy;
  const float* bias;
  int B, H, W, tiles_y, tiles_x;
  int accumulate;
  float* bn_part;                // [tile * WP + pixel group][3][C] or nullptr
  int* bn_cnt;                   // [tile * WP + pixel group] valid pixels
  unsigned* out_rec;             // amax record of what the epilogue stores (max|y|, or max|g| of the BQ epilogue), or nullptr
  // backward-data launches whose output feeds the backward of relu(bn(q)) (catseg_dconv3_bnbwd): see dconv3_b3.hip
  const float* bq_y;
  int bq_ldy;
  const float* bq_stats;
  const float* bq_gamma;
  const float* bq_beta;
  float* bq_part;                // [tile * WP + pixel group][2][C]
};

__device__ __forceinline__ float pl_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

#define PL_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define PL_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))

// wait until all but the n youngest vector-memory operations of this wave have completed (n wave-uniform, 0 .. 8)
__device__ __forceinline__ void pl_wait_vm(int n) {
  switch (n) {
    case 0: PL_WAIT_VM(0); break;
    case 1: PL_WAIT_VM(1); break;
    case 2: PL_WAIT_VM(2); break;
    case 3: PL_WAIT_VM(3); break;
    case 4: PL_WAIT_VM(4); break;
    case 5: PL_WAIT_VM(5); break;
    case 6: PL_WAIT_VM(6); break;
    case 7: PL_WAIT_VM(7); break;
    default: PL_WAIT_VM(8); break;
  }
}

template <class G, bool BQ>
__global__ __launch_bounds__(G::NTHR, G::MINW) void dconv3_pl_kernel(const PlArgs a) {
  __shared__ __attribute__((aligned(256))) unsigned char smem[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = blockIdx.y;
  const int ntile_all = a.B * a.tiles_y * a.tiles_x;
  const int ntile = (ntile_all + G::NT2 - 1) / G::NT2;          // units of NT2 tiles: unit u = tiles NT2 u .. NT2 u + NT2 - 1
  int t_begin, t_end;
  {   // the blocks that share an XCD (equal blockIdx.x % 8) get neighbouring runs of tiles
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    t_begin = (int)((long long)v * ntile / nb);
    t_end = (int)((long long)(v + 1) * ntile / nb);
  }
  if (t_begin >= t_end) return;
  const int total = (t_end - t_begin) * G::TSTEPS;          // K-steps of this block

  if (wave >= G::NCW) {
    // ================================================== helper role: every LDS-DMA of the block ==================================================
    const int hw = wave - G::NCW;
    // halo slot of this lane in the j-th instruction of a (plane, group): slot = 64 j + lane -> (row, column) of the halo tile
    int rel[G::NJ], hr[G::NJ], hc[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
      const int slot = j * 64 + lane;
      const int r = slot / G::HW, c = slot - r * G::HW;
      hr[j] = slot < G::NPX ? r - 1 : -(1 << 20);             // (slots past the halo tile: never valid)
      hc[j] = c - 1;
      rel[j] = ((r - 1) * a.W + (c - 1)) * 16;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.planes, (short)0, (int)(2u * G::NG * a.P16), 0x00020000);
    // the weights of a K-step are WSTEP bytes: every helper streams its quarter (Q bytes) in NW instructions of 1 KB, the last one with
    // the lanes past the quarter switched off -- the same count for every helper, so that the counted waits are compile-time constants
    constexpr int Q = G::WSTEP / 4, NW = (Q + 1023) / 1024;
    const unsigned char* wsrc = (const unsigned char*)a.wimg + (long long)cob * G::TSTEPS * G::WSTEP + hw * Q + lane * 16;
    auto wfill = [&](int q, int slot) {
#ifndef PL_NO_WDMA
      unsigned char* dst = smem + G::NXB * G::RING + slot * G::WSTEP + hw * Q;
      const unsigned char* src = wsrc + (long long)q * G::WSTEP;
#pragma unroll
      for (int i = 0; i < NW; ++i)
        if (i * 1024 + lane * 16 < Q)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                           (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
#endif
    };
    // ---- the NEXT sub-chunk to fetch: (tile, weight chunk, sub-chunk) -> image offset, tile origin, first channel group, ring buffer
    int f_tile = t_begin, f_chunk = 0, f_buf = 0;
    int f_y0[G::NT2], f_x0[G::NT2], f_org[G::NT2];
    unsigned f_img[G::NT2];
    auto f_geom = [&]() {
#pragma unroll
      for (int h = 0; h < G::NT2; ++h) {
        int t = f_tile * G::NT2 + h;
        if (t >= ntile_all) t = ntile_all - 1;              // (an odd tile count: the last unit fetches its one tile twice)
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
        f_y0[h] = ty * G::TH;
        f_x0[h] = tx * G::TW;
        f_org[h] = (f_y0[h] * a.W + f_x0[h]) * 16;
        f_img[h] = (unsigned)(b * a.H * a.W) * 16u;
      }
    };
    f_geom();
    // item m of sub-chunk type U for this helper: U = 0 (4 groups): group = hw, plane = m & 1, j = m >> 1;  U = 1 (2 groups): group = hw & 1,
    // plane = hw >> 1, j = m  (j is a compile-time constant either way: rel[] / hr[] / hc[] are indexed statically)
    auto xissue = [&](auto U_, auto MM_) {
      constexpr int U = decltype(U_)::value, MM = decltype(MM_)::value;
      constexpr int TH_ = MM / G::tile_items(U) < G::NT2 ? MM / G::tile_items(U) : 0, M = MM % G::tile_items(U);     // (tile of the unit, item)
      if constexpr (MM / G::tile_items(U) >= G::NT2) return;
      constexpr int j = (U == 0 ? (M >> 1) : M) < G::NJ ? (U == 0 ? (M >> 1) : M) : 0;
      if constexpr ((U == 0 ? (M >> 1) : M) >= G::NJ) return;
      const int plane = U == 0 ? (M & 1) : (hw >> 1);
      const int lg = U == 0 ? hw : (hw & 1);
      const int g0 = f_chunk * (G::KC / 8) + (U == 0 ? 0 : 4);
      const bool ok = (unsigned)(f_y0[TH_] + hr[j]) < (unsigned)a.H && (unsigned)(f_x0[TH_] + hc[j]) < (unsigned)a.W;
      const unsigned voff = ok ? (unsigned)(f_org[TH_] + rel[j]) : 0xFFFFFFF0u;
      const unsigned soff = (unsigned)(plane * G::NG + g0 + lg) * a.P16 + f_img[TH_];
      unsigned char* dst = smem + f_buf * G::RING + TH_ * G::XBUF + plane * G::XPS + lg * G::KGS + j * 1024;
      if (j * 64 + lane < G::HP)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
    };
    // advance the fetch target past sub-chunk type U; returns false when the block's stream ends
    auto f_advance = [&](auto U_) {
      constexpr int U = decltype(U_)::value;
      f_buf ^= 1;
      if (U + 1 < G::NSUB) return true;                        // (the second sub-chunk of the same weight chunk)
      if (++f_chunk < G::NCHUNK) return true;
      f_chunk = 0;
      // (past the block's last tile the stream is CLAMPED: the last tile is fetched again into ring buffers nobody reads any more, so
      //  that the steady-state K-step has no tail branches and its vmcnt waits are constants)
      if (++f_tile >= t_end) f_tile = t_end - 1;
      f_geom();
      return true;
    };
    // ---- prologue: the first sub-chunk's image and the weights of K-steps 0, 1, 2
#pragma unroll
    for (int m = 0; m < G::sub_items(0); ++m) {
      auto issue = [&](auto M_) { xissue(std::integral_constant<int, 0>{}, M_); };
      // (static unrolling over the item index)
      switch (m) {
        case 0: issue(std::integral_constant<int, 0>{}); break;
        case 1: issue(std::integral_constant<int, 1>{}); break;
        case 2: issue(std::integral_constant<int, 2>{}); break;
        case 3: issue(std::integral_constant<int, 3>{}); break;
        case 4: issue(std::integral_constant<int, 4>{}); break;
        case 5: issue(std::integral_constant<int, 5>{}); break;
        case 6: issue(std::integral_constant<int, 6>{}); break;
        case 7: issue(std::integral_constant<int, 7>{}); break;
        case 8: issue(std::integral_constant<int, 8>{}); break;
        case 9: issue(std::integral_constant<int, 9>{}); break;
        case 10: issue(std::integral_constant<int, 10>{}); break;
        default: issue(std::integral_constant<int, 11>{}); break;
      }
    }
    f_advance(std::integral_constant<int, 0>{});
#pragma unroll
    for (int i = 0; i < G::NWB; ++i) wfill(i % G::TSTEPS, i);
    PL_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();          // P: image 0 and the first weights have landed
    __builtin_amdgcn_s_barrier();          // P2: the compute waves hold their fragments of K-step 0 (slot 0 may be refilled)
    int q3 = G::NWB % G::TSTEPS, s3 = 0;   // step-in-tile index of the weights of step gs + NWB, and their slot (= gs % NWB)
#ifdef PL_WSTAGGER      // (timing-only: every block streams the weight image from a different K-step -- WRONG results)
    q3 = (q3 + blockIdx.x * 5) % G::TSTEPS;
#endif

    // one sub-chunk of type U: NS K-steps; the NEXT sub-chunk (type UN) is fetched during steps 0 .. NS - 3
    auto run_sub = [&](auto U_) {
      constexpr int U = decltype(U_)::value, UN = (U + 1) % G::NSUB;
      constexpr int NS = G::sub_steps(U), NI = G::sub_items(UN), PER = (NI + NS - 3) / (NS - 2);
      auto one_step = [&](auto LS_) {
        constexpr int LS = decltype(LS_)::value;
        constexpr bool X0 = LS * PER < NI && LS <= NS - 3, X1 = PER > 1 && LS * PER + 1 < NI && LS <= NS - 3;
        static_assert(PER <= 2, "at most two halo pieces per helper and K-step");
        wfill(q3, s3);                                      // weights of step gs + NWB (past the end: clamped, see f_advance)
        q3 = q3 + 1 == G::TSTEPS ? 0 : q3 + 1;
#ifndef PL_NO_XDMA
        if constexpr (X0) xissue(std::integral_constant<int, UN>{}, std::integral_constant<int, LS * PER>{});
        if constexpr (X1) xissue(std::integral_constant<int, UN>{}, std::integral_constant<int, LS * PER + 1>{});
#endif
        s3 = s3 == G::NWB - 1 ? 0 : s3 + 1;
        // The weights of step gs + 2 (issued NWB - 2 steps ago) have landed: at most the NW (NWB - 2) youngest operations stay in flight
        // (this and the previous NWB - 3 steps issued at least that many -- halo pieces on top make the wait only stricter).  Behind
        // step NS - 2 the whole next image must be there: everything older than this step's own issues.
#ifndef PL_NO_HWAIT
#if defined(PL_NO_WDMA) || defined(PL_NO_XDMA)
        PL_WAIT_VM(0);
#else
        if constexpr (LS == NS - 2) PL_WAIT_VM(NW + (X0 ? 1 : 0) + (X1 ? 1 : 0));
        else PL_WAIT_VM(NW * (G::NWB - 2));
#endif
#endif
#ifndef PL_NO_LOOPBAR
        __builtin_amdgcn_s_barrier();
#endif
      };
      one_step(std::integral_constant<int, 0>{});
      one_step(std::integral_constant<int, 1>{});
      one_step(std::integral_constant<int, 2>{});
      one_step(std::integral_constant<int, 3>{});
      one_step(std::integral_constant<int, 4>{});
      if constexpr (NS > 5) {
        one_step(std::integral_constant<int, 5>{});
        one_step(std::integral_constant<int, 6>{});
        one_step(std::integral_constant<int, 7>{});
        one_step(std::integral_constant<int, 8>{});
      }
      f_advance(std::integral_constant<int, UN>{});
    };
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
#pragma unroll 1
      for (int chunk = 0; chunk < G::NCHUNK; ++chunk) {
        run_sub(std::integral_constant<int, 0>{});
        if constexpr (G::NSUB == 2) run_sub(std::integral_constant<int, 1>{});
      }
    }
    return;
  }

  // ==================================================== compute role: fragment reads + MFMAs ====================================================
  const int i16 = lane & 15, kg = lane >> 4;
  const int half = wave >> 2, w4 = wave & 3;     // (tile of the unit this wave works on; wave within the tile's four)
  const int wc = w4 / G::WP, wp = w4 % G::WP;
#ifndef PL_NO_SCALE
  const int ex_x = __builtin_amdgcn_readfirstlane(a.x_rec[CS_REC_EXP]);
  const int ex_w = __builtin_amdgcn_readfirstlane(a.w_rec[1]);
#endif
  auto prow = [&](int pt) { return (wp * G::PB + pt) / G::TPW; };
  auto pcol = [&](int pt) { return ((wp * G::PB + pt) % G::TPW) * 16 + i16; };
  int xa[G::PB], xs1[G::PB];     // lane's fragment address in an image: groups 0-3 by kg (steps 0-8) / group kg & 1 (steps 9-13)
#pragma unroll
  for (int pt = 0; pt < G::PB; ++pt) {
    const int px = (prow(pt) * G::HW + pcol(pt)) * 16;
    xa[pt] = half * G::XBUF + kg * G::KGS + px;
    xs1[pt] = half * G::XBUF + (kg & 1) * G::KGS + px;
  }
  const int wb = G::NXB * G::RING + kg * (G::NT * 16) + (wc * G::CB * 16 + i16) * 16;
  h8 xf[G::PB][2], wf[G::CB][2];
  int xbase = 0;                 // ring buffer of the sub-chunk being computed (byte offset)
  // pixel fragments of weight-chunk step WS from the image at byte offset `base`
  auto xread1 = [&](auto WS_, const int base, const int pt) {
    constexpr int WS = decltype(WS_)::value;
    int o;
    if constexpr (WS < 9) {
      o = base + xa[pt] + ((WS / 3) * G::HW + WS % 3) * 16;
    } else {
      constexpr PlUnit ua = pl_unit_of(WS, 0), ub = pl_unit_of(WS, 1);
      constexpr int offa = ((ua.tap / 3) * G::HW + ua.tap % 3) * 16, offb = ((ub.tap / 3) * G::HW + ub.tap % 3) * 16;
      o = base + xs1[pt] + (kg >> 1 ? offb : offa);
    }
    xf[pt][0] = *(const h8*)(smem + o);
    xf[pt][1] = *(const h8*)(smem + o + G::XPS);
  };
  auto wread1 = [&](const int slot, const int ct) {
    const unsigned char* wbuf = smem + wb + slot * G::WSTEP + ct * 256;
    wf[ct][0] = *(const h8*)(wbuf);
    wf[ct][1] = *(const h8*)(wbuf + G::WPS);
  };
  f32x4 acc[G::CB][G::PB];
  int s3n = 1;                   // slot of the NEXT K-step's weights
  __builtin_amdgcn_s_barrier();  // P
#pragma unroll
  for (int ct = 0; ct < G::CB; ++ct) wread1(0, ct);
#pragma unroll
  for (int pt = 0; pt < G::PB; ++pt) xread1(std::integral_constant<int, 0>{}, 0, pt);
  PL_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();  // P2

  // K-steps of one sub-chunk of type U (weight-chunk steps WS0 .. WS0 + NS - 1); behind its last MFMAs the fragments of the NEXT sub-chunk's
  // first step (type UN, the other ring buffer) roll in
  auto run_sub = [&](auto U_) {
    constexpr int U = decltype(U_)::value, UN = (U + 1) % G::NSUB;
    constexpr int NS = G::sub_steps(U), WS0 = U == 0 ? 0 : 9, WSN = UN == 0 ? 0 : 9;
    auto one_step = [&](auto LS_) {
      constexpr int LS = decltype(LS_)::value;
      // MFMA order: pixel tile outer, output-channel tile inner -- a fragment set is re-read for the next K-step right behind its last use,
      // and the next step then needs the sets in the order they were re-read (xf[0] at mid-step, wf[0], wf[1], ... during the second
      // pixel tile, xf[PB - 1] and wf[CB - 1] last: they are wanted 6 - 9 MFMAs into the next step).  No s_waitcnt in front of the
      // barrier: a fragment read issued in step s is complete before the wave's own MFMAs of step s + 1 consume it, and the helpers
      // overwrite an LDS region at the earliest one barrier after the step that read it last.
#pragma unroll
      for (int pt = 0; pt < G::PB; ++pt) {
#pragma unroll
        for (int ct = 0; ct < G::CB; ++ct) {
          f32x4 c = acc[ct][pt];
#ifdef PL_NO_MFMA       // (timing-only: one product instead of three)
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct][0], xf[pt][1] + xf[pt][0], c, 0, 0, 0);
#else
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct][0], xf[pt][1], c, 0, 0, 0);   // h l
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct][1], xf[pt][0], c, 0, 0, 0);   // l h
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ct][0], xf[pt][0], c, 0, 0, 0);   // h h
#endif
          acc[ct][pt] = c;
#ifndef PL_NO_DSREAD
          if (pt == G::PB - 1) {                    // the next step's weight fragments behind the last use of these registers
            __builtin_amdgcn_sched_barrier(0);
            wread1(s3n, ct);
            __builtin_amdgcn_sched_barrier(0);
          }
#endif
        }
#ifndef PL_NO_DSREAD
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LS + 1 < NS) xread1(std::integral_constant<int, WS0 + LS + 1>{}, xbase, pt);
        else xread1(std::integral_constant<int, WSN>{}, xbase ^ G::RING, pt);
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      s3n = s3n == G::NWB - 1 ? 0 : s3n + 1;
      PL_WAIT_LGKM0();      // every fragment read of this step has returned before the barrier lets the helpers refill what it read

#ifndef PL_NO_LOOPBAR
      __builtin_amdgcn_s_barrier();
#endif
    };
    one_step(std::integral_constant<int, 0>{});
    one_step(std::integral_constant<int, 1>{});
    one_step(std::integral_constant<int, 2>{});
    one_step(std::integral_constant<int, 3>{});
    one_step(std::integral_constant<int, 4>{});
    if constexpr (NS > 5) {
      one_step(std::integral_constant<int, 5>{});
      one_step(std::integral_constant<int, 6>{});
      one_step(std::integral_constant<int, 7>{});
      one_step(std::integral_constant<int, 8>{});
    }
    xbase ^= G::RING;
  };

#pragma unroll 1
  for (int unit = t_begin; unit < t_end; ++unit) {
    const int tile_raw = unit * G::NT2 + half;
    const bool t_ok = tile_raw < ntile_all;                  // (the second tile of the last unit of an odd tile count does not exist)
    const int tile = t_ok ? tile_raw : ntile_all - 1;
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const long long img0 = (long long)b * a.H * a.W;
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
      for (int pt = 0; pt < G::PB; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int chunk = 0; chunk < G::NCHUNK; ++chunk) {
      run_sub(std::integral_constant<int, 0>{});
      if constexpr (G::NSUB == 2) run_sub(std::integral_constant<int, 1>{});
    }

    // ---- epilogue (no LDS, no barrier: the helpers keep prefetching the next tile) -----------------------------------------------------
#ifndef PL_NO_SCALE
    if (ex_x + ex_w >= -120 && ex_x + ex_w <= 120) {     // back to the operands' scale: one exact multiplication by 2^-(e_x + e_w)
      const float sc = __builtin_ldexpf(1.f, -(ex_x + ex_w));
#pragma unroll
      for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
        for (int pt = 0; pt < G::PB; ++pt) acc[ct][pt] *= sc;
    } else {
#pragma unroll
      for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
        for (int pt = 0; pt < G::PB; ++pt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ct][pt][r] = __builtin_ldexpf(__builtin_ldexpf(acc[ct][pt][r], -ex_x), -ex_w);
    }
#endif
    bool p_ok[G::PB];
    int nvalid = 0;
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) {
      p_ok[pt] = t_ok && y0 + prow(pt) < a.H && x0 + pcol(pt) < a.W;
      nvalid += __builtin_popcountll(__builtin_amdgcn_ballot_w64(p_ok[pt]) & 0xFFFFull);
    }
    const int cl = wc * G::CB * 16 + 4 * kg;              // block-local channel of (ct = 0, r = 0)
    const int co0 = cob * G::NT + cl;
    const int prw = tile * G::WP + wp;                    // this wave's row of the partial buffers
    unsigned amax = 0;
    if constexpr (BQ) {
      // g = acc where relu(bn(q)) > 0 (z recomputed exactly as bn_apply evaluates it), stored; per (wave, channel) sums of g and g * xhat
      float* part = a.bq_part + (long long)prw * 2 * G::C + cob * G::NT;
#pragma unroll
      for (int ct = 0; ct < G::CB; ++ct) {
        const int c = co0 + ct * 16;
        const f32x4 mean = *(const f32x4*)(a.bq_stats + c), inv = *(const f32x4*)(a.bq_stats + G::C + c);
        const f32x4 sc = *(const f32x4*)(a.bq_gamma + c) * inv;   // scale exactly as bn_finalize_kernel stored it
        const f32x4 be = *(const f32x4*)(a.bq_beta + c);
        f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pt = 0; pt < G::PB; ++pt) {
          const long long px = img0 + (long long)(y0 + prow(pt)) * a.W + x0 + pcol(pt);
          const f32x4 q = p_ok[pt] ? *(const f32x4*)(a.bq_y + px * a.bq_ldy + c) : f32x4{0.f, 0.f, 0.f, 0.f};
          f32x4 g;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = __builtin_fmaf(q[r] - mean[r], sc[r], be[r]);   // = bn_affine (norm.hip), the forward's own expression
            g[r] = (p_ok[pt] && z > 0.f) ? acc[ct][pt][r] : 0.f;
          }
          if (p_ok[pt]) *(f32x4*)(a.y + px * a.ldy + c) = g;
          amax = max(amax, cs_abs_bits4(g));
          sg += g;
          sgx += g * ((q - mean) * inv);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v1 = pl_row16_sum(sg[r]), v2 = pl_row16_sum(sgx[r]);
          if (i16 == 0 && t_ok) {
            part[cl + ct * 16 + r] = v1;
            part[G::C + cl + ct * 16 + r] = v2;
          }
        }
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < G::CB; ++ct) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) bv = *(const f32x4*)(a.bias + co0 + ct * 16);
#pragma unroll
        for (int pt = 0; pt < G::PB; ++pt) {
          acc[ct][pt] += bv;
#ifdef PL_NO_STORE
          if (p_ok[pt] && acc[ct][pt][0] == 123.456f) {
#else
          if (p_ok[pt]) {
#endif
            float* dst = a.y + (img0 + (long long)(y0 + prow(pt)) * a.W + x0 + pcol(pt)) * a.ldy + co0 + ct * 16;
            f32x4 v = acc[ct][pt];
            if (a.accumulate) v += *(const f32x4*)dst;
            *(f32x4*)dst = v;
            amax = max(amax, cs_abs_bits4(v));
          }
        }
      }
      if (a.bn_part) {
        // (K, sum(v - K), sum((v - K)^2)) per channel over this wave's valid pixels, ONE pass: the shift K is the value of the wave's first
        // pixel (lane 0 of each 16-lane row holds it for that row's four channels: DPP row_share; it is a valid pixel whenever the wave
        // has any: rows run top to bottom, columns left to right) -- any sample of the data is as good a shift as the mean, and
        // catseg_bn_finalize_counts merges (K, s1, s2) rows with arbitrary K.  The 16 lanes of a k-group hold 16 pixels of the same channels.
        float* part = a.bn_part + (long long)prw * 3 * G::C + cob * G::NT;
#pragma unroll
        for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (through a scalar copy: __builtin_bit_cast applied to the vector ELEMENT acc[ct][0][r] itself reads element 0 for every r -- hipcc 7.2;
            //  until round 6 the shift of channels r = 1 .. 3 of a quad was therefore channel r = 0's first pixel: a valid shift, but not this channel's)
            const float a0 = acc[ct][0][r];
            const float K = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x150, 0xF, 0xF, true));
            float d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int pt = 0; pt < G::PB; ++pt) {
              const float d = p_ok[pt] ? acc[ct][pt][r] - K : 0.f;
              d1 += d;
              d2 += d * d;
            }
            d1 = pl_row16_sum(d1);
            d2 = pl_row16_sum(d2);
            if (i16 == 0 && t_ok) {
              const int c = cl + ct * 16 + r;
              part[c] = nvalid > 0 ? K : 0.f;
              part[G::C + c] = d1;
              part[2 * G::C + c] = d2;
            }
          }
        if (lane == 0 && cob == 0 && wc == 0 && t_ok) a.bn_cnt[prw] = nvalid;
      }
    }
    if (a.out_rec) {     // max |stored value| of this wave -> one of the record's 16 slots (order-independent: deterministic)
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o));
      if (lane == 0 && amax) atomicMax(a.out_rec + ((blockIdx.x * 4 + wave) % CS_AMAX_SLOTS) * CS_AMAX_STRIDE, amax);
    }
  }
}

// standalone producer of the planes (tests, and tensors whose producer writes no planes): exponent from the record's max|x| words (the
// caller has filled them: catseg_amax / a producer's epilogue), planes as csrc/planes.h lays them out
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const float* __restrict__ x, int ld, long long rows, int C, unsigned* __restrict__ rec,
                                                              unsigned char* __restrict__ planes) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[CsPlaneTile::BYTES];
  const unsigned bound = cs_amax_read(rec);
  const int e = cs_plane_exponent(bound);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ((int*)rec)[CS_REC_EXP] = e;
    rec[CS_REC_FINAL] = bound;
  }
  const float sc = __builtin_ldexpf(1.f, e);
  const int NG = C >> 3, gblocks = (NG + 7) >> 3;
  const long long rblocks = (rows + CsPlaneTile::ROWS - 1) / CsPlaneTile::ROWS;          // 8 channel groups x 32 rows per pass
  for (long long t = blockIdx.x; t < rblocks * gblocks; t += gridDim.x) {
    const long long row0 = (t / gblocks) * CsPlaneTile::ROWS;
    const int g0 = (int)(t % gblocks) << 3, ng = NG - g0 < 8 ? NG - g0 : 8;
    for (int i = threadIdx.x; i < CsPlaneTile::ROWS * ng; i += 256) {
      const int row = i / ng, g = i - row * ng;
      if (row0 + row < rows) {
        const float* src = x + (row0 + row) * ld + (g0 + g) * 8;
        const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
        const float xs[8] = {v0[0] * sc, v0[1] * sc, v0[2] * sc, v0[3] * sc, v1[0] * sc, v1[1] * sc, v1[2] * sc, v1[3] * sc};
        CsPlaneTile::stage(sm, row, g, xs);
      }
    }
    __syncthreads();
    const long long left = rows - row0;
    CsPlaneTile::flush(sm, planes, rows, NG, row0, left < CsPlaneTile::ROWS ? (int)left : CsPlaneTile::ROWS, g0, ng);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void amax_f32_kernel(const float* __restrict__ x, int ld, long long rows, int C, unsigned* __restrict__ rec) {
  const int cpt = C >> 2;
  unsigned m = 0;
  CS_QUAD_LOOP(rows, cpt, r, c) {
    m = max(m, cs_abs_bits4(*(const f32x4*)(x + r * ld + c)));
  }
  cs_amax_commit(m, rec);
}

using Pl48 = PlCfg<48, 48, 1, 4, 2, 8, 1>;     // tile 8 x 16, wave = 48 co x 32 px
using Pl64 = PlCfg<64, 64, 2, 2, 2, 4, 1>;     // tile 4 x 16, wave = 32 co x 32 px (the stage-1 bottlenecks' 3x3)
using Pl96 = PlCfg<96, 96, 2, 2, 2, 4, 1>;     // tile 4 x 16, wave = 48 co x 32 px
using Pl192 = PlCfg<192, 96, 2, 2, 2, 2, 2>;   // tile 2 x 32, two co blocks
using Pl384 = PlCfg<384, 96, 2, 2, 2, 2, 2>;   // four co blocks
// the same tiles, two per block: eight compute waves share one stream of weights (half the LDS-DMA bytes of the weights per output), one block per CU
using Pl96P = PlCfg<96, 96, 2, 2, 2, 4, 1, 2>;
using Pl192P = PlCfg<192, 96, 2, 2, 2, 2, 2, 2>;
using Pl384P = PlCfg<384, 96, 2, 2, 2, 2, 2, 2>;

struct PlPlan { int kind, NT, TH, TW, WP; };
PlPlan pl_plan(int C) {
  if (C == 48) return {1, Pl48::NT, Pl48::TH, Pl48::TW, Pl48::WP};
  if (C == 64) return {2, Pl64::NT, Pl64::TH, Pl64::TW, Pl64::WP};
  if (C == 96) return {3, Pl96::NT, Pl96::TH, Pl96::TW, Pl96::WP};
  if (C == 192) return {4, Pl192::NT, Pl192::TH, Pl192::TW, Pl192::WP};
  if (C == 384) return {5, Pl384::NT, Pl384::TH, Pl384::TW, Pl384::WP};
  return {0, 0, 0, 0, 0};
}

template <class G>
void pl_launch(const PlArgs& a, int C, hipStream_t st) {
  const int ntile = (a.B * a.tiles_y * a.tiles_x + G::NT2 - 1) / G::NT2;        // units of NT2 tiles
  const int slots = catseg_g_pl_slots / G::NT2;      // two blocks per CU, or one of twice the size (catseg_debug_set_dconv3_pl_slots: tuning runs)
  // a block per tile while the tiles fit the block slots about once or twice; beyond that persistent blocks walking runs of tiles
  const int nb = ntile * (C / G::NT) > 2 * slots ? slots / (C / G::NT) : ntile;
  if (a.bq_part) hipLaunchKernelGGL((dconv3_pl_kernel<G, true>), dim3(nb, C / G::NT), dim3(G::NTHR), 0, st, a);
  else hipLaunchKernelGGL((dconv3_pl_kernel<G, false>), dim3(nb, C / G::NT), dim3(G::NTHR), 0, st, a);
}

int pl_run(int B, int H, int W, int C, const void* planes, const void* x_rec, const void* wimg, const void* w_rec, const float* bias, float* y,
           int ldy, int accumulate, float* bn_part, size_t bn_part_floats, int* bn_counts, void* out_rec, const float* bq_y, int bq_ldy,
           const float* bq_stats, const float* bq_gamma, const float* bq_beta, float* bq_part, size_t bq_part_floats, catseg_stream_t stream) {
  const PlPlan p = pl_plan(C);
  CS_REQUIRE(p.kind, "dconv3 (planes): unsupported channel count %d", C);
  CS_REQUIRE(B > 0 && H > 0 && W > 0 && planes && x_rec && wimg && w_rec && y, "dconv3 (planes): bad args");
  CS_REQUIRE(ldy >= C && ldy % 4 == 0 && cs_aligned16(planes) && cs_aligned16(y) && cs_aligned16(wimg) && cs_aligned16(bias),
             "dconv3 (planes): alignment / row strides");
  const long long P = (long long)B * H * W;
  CS_REQUIRE(4 * P * C < (1LL << 32) && (long long)H * W * ldy < (1LL << 31), "dconv3 (planes): tensor too large for 32-bit offsets");
  PlArgs a;
  a.planes = (const unsigned char*)planes; a.P16 = (unsigned)(P * 16); a.x_rec = (const int*)x_rec;
  a.wimg = (const u16*)wimg; a.w_rec = (const int*)w_rec; a.y = y; a.ldy = ldy; a.bias = bias;
  a.B = B; a.H = H; a.W = W;
  a.tiles_y = (H + p.TH - 1) / p.TH;
  a.tiles_x = (W + p.TW - 1) / p.TW;
  a.accumulate = accumulate;
  a.bn_part = bn_part; a.bn_cnt = bn_counts; a.out_rec = (unsigned*)out_rec;
  a.bq_y = bq_y; a.bq_ldy = bq_ldy; a.bq_stats = bq_stats; a.bq_gamma = bq_gamma; a.bq_beta = bq_beta; a.bq_part = bq_part;
  const long long nrow = (long long)B * a.tiles_y * a.tiles_x * p.WP;
  if (bn_part) CS_REQUIRE(bn_counts && bn_part_floats >= (size_t)nrow * 3 * C, "dconv3 (planes): BatchNorm partial buffer too small");
  if (bq_part) CS_REQUIRE(bq_part_floats >= (size_t)nrow * 2 * C, "dconv3 bnbwd (planes): partial buffer too small");
  hipStream_t st = (hipStream_t)stream;
  const int pair = catseg_g_pl_pair;
  if (p.kind == 1) pl_launch<Pl48>(a, C, st);       // (two tiles of 48 channels would need four halo pieces per helper and K-step)
  else if (p.kind == 2) pl_launch<Pl64>(a, C, st);
  else if (p.kind == 3) { if (pair & 2) pl_launch<Pl96P>(a, C, st); else pl_launch<Pl96>(a, C, st); }
  else if (p.kind == 4) { if (pair & 4) pl_launch<Pl192P>(a, C, st); else pl_launch<Pl192>(a, C, st); }
  else { if (pair & 8) pl_launch<Pl384P>(a, C, st); else pl_launch<Pl384>(a, C, st); }
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

}  // namespace

extern "C" int catseg_dconv3_pl_supported(int C) { return pl_plan(C).kind != 0; }

// blocks of the planes kernel the runtime places on one CU (hipOccupancyMaxActiveBlocksPerMultiprocessor): the two-blocks-per-CU design of the
// default form is a property of the LDS and register footprint, checked by tests/test_dconv3_pl_gpu.py.  pair != 0: the two-tiles-per-block form.
extern "C" int catseg_debug_dconv3_pl_occupancy(int C, int pair) {
  int n = -1;
  hipError_t e = hipErrorInvalidValue;
  if (C == 48) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dconv3_pl_kernel<Pl48, false>, Pl48::NTHR, 0);
  else if (C == 64) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dconv3_pl_kernel<Pl64, false>, Pl64::NTHR, 0);
  else if (C == 96) e = pair ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dconv3_pl_kernel<Pl96P, false>, Pl96P::NTHR, 0)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dconv3_pl_kernel<Pl96, false>, Pl96::NTHR, 0);
  else if (C == 192) e = pair ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dconv3_pl_kernel<Pl192P, false>, Pl192P::NTHR, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dconv3_pl_kernel<Pl192, false>, Pl192::NTHR, 0);
  else if (C == 384) e = pair ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dconv3_pl_kernel<Pl384P, false>, Pl384P::NTHR, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dconv3_pl_kernel<Pl384, false>, Pl384::NTHR, 0);
  return e == hipSuccess ? n : -1;
}

// rows of the per-wave BatchNorm partial buffers of a launch: tiles x pixel groups per tile
extern "C" int catseg_dconv3_pl_rows(int C, int B, int H, int W) {
  const PlPlan p = pl_plan(C);
  if (!p.kind) return 0;
  return B * ((H + p.TH - 1) / p.TH) * ((W + p.TW - 1) / p.TW) * p.WP;
}

extern "C" size_t catseg_planes_bytes(long long rows, int C) { return (size_t)rows * (size_t)C * 4; }

// x [rows][ld] fp32 -> planes; record: the tensor's amax record (include/catseg.h: CATSEG_AMAX_RECORD_BYTES).  compute_amax != 0: the
// record is zeroed and max|x| taken here first (two more launches); otherwise its max|x| words are taken as they are (a larger value is
// safe).  Writes the exponent into word 1 of the record.
extern "C" int catseg_planes_from_f32(const float* x, int ld, long long rows, int C, void* record, void* planes, int compute_amax,
                                      catseg_stream_t stream) {
  CS_REQUIRE(x && record && planes && rows > 0 && C > 0 && C % 8 == 0 && ld >= C && ld % 4 == 0 && cs_aligned16(x) && cs_aligned16(planes),
             "planes_from_f32: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (compute_amax) {
    if (hipMemsetAsync(record, 0, CS_AMAX_WORDS * 4, st) != hipSuccess) { catseg_set_error("planes_from_f32: memset failed"); return CATSEG_EHIP; }
    long long blocks = (rows * (C / 4) + 255) / 256;
    hipLaunchKernelGGL(amax_f32_kernel, dim3((int)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, st, x, ld, rows, C, (unsigned*)record);
  }
  const long long tiles = ((rows + 127) / 128) * ((C / 8 + 7) / 8);
  hipLaunchKernelGGL(planes_from_f32_kernel, dim3((int)(tiles > 4096 ? 4096 : tiles)), dim3(256), 0, st, x, ld, rows, C, (unsigned*)record,
                     (unsigned char*)planes);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_dconv3_f16x2 on producer-written planes (csrc/planes.h).  planes_record: word 1 = the exponent of the planes.  bn_part: per-WAVE
// rows ([catseg_dconv3_pl_rows][3][C] + bn_counts[rows]) for catseg_bn_finalize_counts.  out_record (may be null): max|y| is folded into
// its 16 amax slots.
extern "C" int catseg_dconv3_pl(int B, int H, int W, int C, const void* planes, const void* planes_record, const void* wimg, const void* w_record,
                                const float* bias, float* y, int ldy, int accumulate, float* bn_part, size_t bn_part_floats, int* bn_counts,
                                void* out_record, catseg_stream_t stream) {
  return pl_run(B, H, W, C, planes, planes_record, wimg, w_record, bias, y, ldy, accumulate, bn_part, bn_part_floats, bn_counts, out_record, nullptr,
                0, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

// catseg_dconv3_bnbwd_f16x2 on planes of dy: g (masked) to `g`, per-wave sums of g and g * xhat to part ([catseg_dconv3_pl_rows][2][C]),
// max|g| to out_record
extern "C" int catseg_dconv3_pl_bnbwd(int B, int H, int W, int C, const void* dy_planes, const void* dy_record, const void* wimg_bwd,
                                      const void* w_record, float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma,
                                      const float* beta, float* part, size_t part_floats, void* out_record, catseg_stream_t stream) {
  CS_REQUIRE(q && stats && gamma && beta && part, "dconv3 bnbwd (planes): bad args");
  CS_REQUIRE(ldq >= C && ldq % 4 == 0 && cs_aligned16(q) && cs_aligned16(stats) && cs_aligned16(gamma) && cs_aligned16(beta) && cs_aligned16(part),
             "dconv3 bnbwd (planes): alignment / row strides");
  CS_REQUIRE((long long)H * W * (long long)ldq < (1LL << 31), "dconv3 bnbwd (planes): image too large for 32-bit offsets");
  return pl_run(B, H, W, C, dy_planes, dy_record, wimg_bwd, w_record, nullptr, g, ldg, 0, nullptr, 0, nullptr, out_record, q, ldq, stats, gamma, beta,
                part, part_floats, stream);
}
