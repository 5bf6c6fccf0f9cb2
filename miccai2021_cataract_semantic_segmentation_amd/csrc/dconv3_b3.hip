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
//
// This file compiles twice.  As it is: three bf16 planes, six products (the text above).  Through dconv3_f16x2.hip (-DDC_H2): TWO fp16
// planes and THREE products, the arithmetic of igemm_f16x2.hip -- the activation is scaled by 2^e (e from the 8-byte amax record its
// PRODUCER filled: catseg_bn_apply_amax / catseg_add_n_act_amax / catseg_bn_backward_amax) while it is split in registers, the weight
// image carries its own exponent, the epilogue scales the accumulators back; entry points catseg_dconv3_f16x2*.
#include "common.h"

#ifdef DC_H2
#define DC_NPL 2
#define DC_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#define dconv3_b3_kernel dconv3_h2_kernel
#define dconv3_b3_spec_kernel dconv3_h2_spec_kernel
#define dconv3_prep_kernel dconv3_h2_prep_kernel
#define dconv3_prep_batch_kernel dconv3_h2_prep_batch_kernel
#elif defined(DC_AB2)
// (timing-only build -DDC_AB2: two planes and three products of the bf16 arithmetic -- WRONG results; what the f16x2 build was worth
//  before it existed: the l plane is neither stored, streamed nor read, the products mm / hl / lh are dropped)
#define DC_NPL 2
#define DC_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#else
#define DC_NPL 3
#define DC_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

namespace {

#ifdef DC_H2
typedef _Float16 dc_t;
#else
typedef __bf16 dc_t;
#endif
typedef dc_t bf16x8 __attribute__((ext_vector_type(8)));     // eight 16-bit plane elements (fp16 in the DC_H2 build)
typedef unsigned short u16;

// prescale exponent from the bits of a tensor's max |x| (igemm_f16x2.hip: h2_exponent): amax * 2^e in [2^14, 2^15)
__host__ __device__ inline int dc_exponent(unsigned amax_bits) {
  const int ex = (int)((amax_bits >> 23) & 0xFF);
  if (ex == 0 || ex == 255) return 0;
  const int e = 14 - (ex - 127);
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}

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

template <int C_, int KC_, int NT_, int WC_, int WP_, int PB_, int TPH_, int TPW_, int PH_, int PW_, bool XPF_, bool SPEC_ = false>
struct DcCfg {
  static constexpr bool SPEC = SPEC_;                     // default variant: 4 compute + 4 helper waves per block (dconv3_b3_spec_kernel)
  static constexpr bool XPF = XPF_;                       // pixel fragments of step t + 1 read under the MFMAs of step t
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
  static constexpr int XBYTES = DC_NPL * XPS;
  static constexpr int WPS = 64 * NT;                      // bytes per plane of one K-step of weights: [4 k-groups][NT][16 B]
  static constexpr int WSTEP = DC_NPL * WPS;
  static constexpr int NCHUNK = C / KC, NSTEP = steps_of(KC);
  // staging: one wave instruction = a unit of 16 consecutive halo pixels x 4 channel groups (lane = pixel + 16 * group): the
  // 8-lane groups of a ds_write_b128 then hit 8 consecutive 16-byte slots (conflict-free; a lane order with the channel groups
  // innermost wrote NKG-way conflicts: 32 % of all LDS cycles)
  static constexpr int NPX = HH * HW;
  static constexpr int NPG = (NPX + 15) / 16, NGB = (NKG + 3) / 4;
  static constexpr int IPT = (NPG * NGB + NW - 1) / NW;    // units per wave
  static constexpr int WITEMS = WSTEP / 1024;              // LDS-DMA wave instructions per K-step
  static_assert(WSTEP % 1024 == 0, "weights of a K-step in whole 1 KB pieces");
  static constexpr int LDS = XBYTES + 2 * WSTEP;
  static constexpr int SCR = 3 * WP * NT * 4;              // epilogue scratch (floats -> bytes), inside the X image
  static_assert(SCR <= XBYTES, "epilogue scratch");
  static_assert(IPT + 2 <= NSTEP, "the prefetch of the next halo tile is spread over the K-steps: item i is loaded in step i, split in step i + 2");
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
  // backward through the ReLU + BatchNorm that produced this convolution's INPUT (backward-data launches, catseg_dconv3_bnbwd):
  // the output is the gradient of z = relu(bn(q)); the epilogue masks it with z > 0 (z recomputed from q exactly as bn_apply_kernel
  // evaluates it) and leaves the per-tile sums of g and g * xhat that bn_bwd_partial_kernel would have needed another pass for
  const float* bq_y;      // q, the pre-normalisation tensor [B][H][W][ld]
  int bq_ldy;
  const float* bq_stats;  // [mean(C), invstd(C)]
  const float* bq_gamma;
  const float* bq_beta;
  float* bq_part;         // [tile][2][C] or nullptr
  // DC_H2: the activation's amax record (word 0 = bits of max|x|, filled by its producer) and the weight image's record (word 1 = exponent)
  const unsigned* x_rec;
  const int* w_rec;
};

__device__ __forceinline__ void dc_glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// sum over the 16 lanes of a DPP row (every lane gets the total): quad xor 1, xor 2, half-row mirror, row mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

// s_waitcnt through the builtin (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]): unlike inline asm the
// compiler's own wait insertion sees these, so it does not re-wait for what they have retired
#define DC_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)
#define DC_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))

// Persistent blocks (two per CU): a block walks a contiguous run of pixel tiles.  Per tile and channel chunk the prefetched fp32
// halo tile is split and written to LDS (stash); then NSTEP K-steps run with ONE raw barrier each.  All operands of K-step t are in
// registers when it starts: during step t a wave issues, in this order, the LDS-DMA of the weights of step t + 2 (into the slot whose
// fragments it already holds), two global loads of the prefetch of the NEXT chunk / tile (spread over the steps, so that the counted
// wait that ends the step -- vmcnt(2): everything but those two loads -- never waits for an HBM access), the LDS reads of the weight
// and pixel fragments of step t + 1, and then the 6 CB PB MFMAs of step t, which depend on none of them.
// Epilogue of a backward-data launch whose output feeds the backward of relu(bn(q)) (see DcArgs::bq_*): g = acc where z > 0,
// stored; per (tile, channel) sum g and sum g * xhat over the tile's valid pixels (DPP row sums over the 16 pixels of a k-group,
// the WP waves of a channel group through LDS), in the [block][2][C] layout bn_bwd_finalize_kernel merges.
template <class G, class PRow, class PCol>
__device__ __forceinline__ void dc_bnbwd_epilogue(const DcArgs& a, f32x4 (&acc)[G::CB][G::PB], const bool (&p_ok)[G::PB], long long img0, int y0,
                                                  int x0, int co0, int cl, int wp, int i16, int tile, int cob, unsigned char* smem,
                                                  PRow prow, PCol pcol) {
  f32x4 q[G::CB][G::PB];
#pragma unroll
  for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) {
      const long long px = img0 + (long long)(y0 + prow(pt)) * a.W + x0 + pcol(pt);
      q[ct][pt] = p_ok[pt] ? *(const f32x4*)(a.bq_y + px * a.bq_ldy + co0 + ct * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  float* scr = (float*)smem;    // [2 WP][NT]  (the K loop's last barrier is behind every LDS read of the image)
#pragma unroll
  for (int ct = 0; ct < G::CB; ++ct) {
    const int c = co0 + ct * 16;
    const f32x4 mean = *(const f32x4*)(a.bq_stats + c), inv = *(const f32x4*)(a.bq_stats + G::C + c);
    const f32x4 sc = *(const f32x4*)(a.bq_gamma + c) * inv;   // scale exactly as bn_finalize_kernel stored it
    const f32x4 be = *(const f32x4*)(a.bq_beta + c);
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) {
      f32x4 g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = __builtin_fmaf(q[ct][pt][r] - mean[r], sc[r], be[r]);   // = bn_affine (norm.hip), the forward's own expression
        g[r] = (p_ok[pt] && z > 0.f) ? acc[ct][pt][r] : 0.f;
      }
      if (p_ok[pt]) *(f32x4*)(a.y + (img0 + (long long)(y0 + prow(pt)) * a.W + x0 + pcol(pt)) * a.ldy + c) = g;
      const f32x4 xh = (q[ct][pt] - mean) * inv;
      sg += g;
      sgx += g * xh;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v1 = row16_sum(sg[r]), v2 = row16_sum(sgx[r]);
      if (i16 == 0) {
        scr[wp * G::NT + cl + ct * 16 + r] = v1;
        scr[(G::WP + wp) * G::NT + cl + ct * 16 + r] = v2;
      }
    }
  }
  DC_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();
  if (wp == 0 && i16 == 0) {
    float* part = a.bq_part + (long long)tile * 2 * G::C + cob * G::NT;
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
        part[cl + ct * 16 + r] = d1;
        part[G::C + cl + ct * 16 + r] = d2;
      }
  }
  DC_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();   // the scratch is read: the next tile's stash may overwrite it
}

template <class G, bool BQ = false>   // BQ: the catseg_dconv3_bnbwd epilogue (a separate instantiation: the plain kernels keep their registers)
__global__ __launch_bounds__(G::NTHR, 2) void dconv3_b3_kernel(const DcArgs a) {
  __shared__ __attribute__((aligned(256))) unsigned char smem[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave / G::WP, wp = wave % G::WP;
  const int i16 = lane & 15, kg = lane >> 4;
  const int cob = blockIdx.y;                       // output-channel block

  // ---- this block's run of tiles; the blocks that share an XCD (equal blockIdx.x % 8) get neighbouring runs ---------------------
  const int ntile = a.B * a.tiles_y * a.tiles_x;
#ifdef DC_H2
  const int ex_x = __builtin_amdgcn_readfirstlane(dc_exponent(cs_amax_read(a.x_rec)));   // prescale of the activation (its producer's amax record)
  const int ex_w = __builtin_amdgcn_readfirstlane(a.w_rec[1]);              // prescale of the weight image
  const float sc_x = __builtin_ldexpf(1.f, ex_x);                            // 2^e_x (|e| <= 100: a normal float; x * 2^e is exact)
#endif
  int t_begin, t_end;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    t_begin = (int)((long long)v * ntile / nb);
    t_end = (int)((long long)(v + 1) * ntile / nb);
  }
  if (t_begin >= t_end) return;
#ifdef DC_PRIO
  // static priority: the two blocks that share a CU (dispatch order: block b and b + grid / 2) do not arbitrate step by step
  if (blockIdx.x < (gridDim.x >> 1)) __builtin_amdgcn_s_setprio(DC_PRIO);
#endif

  // ---- staging: unit u = wave + i NW covers 16 halo pixels x 4 channel groups, lane = pixel + 16 group ------------------------
  // Prefetch of the next (tile, chunk): raw buffer loads over ONE image (rows above / below it are out of range: the hardware
  // returns zeros = the convolution's padding; columns left / right of it get the out-of-range marker), item i in K-step i;
  // two steps later the loaded fp32 values are split into the three bf16 planes (VALU work in the shadow of that step's MFMAs).
  auto item_px = [&](int i) { return ((wave + i * G::NW) / G::NGB) * 16 + i16; };
  auto item_g8 = [&](int i) { return ((wave + i * G::NW) % G::NGB) * 4 + kg; };
  auto item_live = [&](int i) { return wave + i * G::NW < G::NPG * G::NGB && item_px(i) < G::NPX && item_g8(i) < G::NKG; };
  int it_off[G::IPT], it_hx[G::IPT];     // byte offset of the item relative to the tile origin pixel (may be negative), halo column - 1
#pragma unroll
  for (int i = 0; i < G::IPT; ++i) {
    const int px = item_px(i);
    it_hx[i] = px % G::HW - 1;
    it_off[i] = (((px / G::HW - 1) * a.W + it_hx[i]) * a.ldx + (item_live(i) ? item_g8(i) * 8 : 0)) * 4;
  }
  f32x4 pre[G::IPT][2];      // (whole-vector bit casts only: __builtin_bit_cast of ONE vector element reads element 0, hipcc 7.2)
  bf16x8 pl[G::IPT][DC_NPL];
  const int img_bytes = ((a.H * a.W - 1) * a.ldx + G::KC) * 4;
  __amdgpu_buffer_rsrc_t f_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, img_bytes, 0x00020000);
  int f_x0 = 0, f_org = 0;           // fetch target: first column of the tile, byte offset of its origin pixel in the image
  auto target = [&](int tile, int chunk) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    f_x0 = tx * G::TW;
    f_org = (ty * G::TH * a.W + f_x0) * a.ldx * 4;
    f_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)b * a.H * a.W * a.ldx + chunk * G::KC), (short)0, img_bytes, 0x00020000);
  };
  auto fetch_item = [&](int i) {
    const bool okx = (unsigned)(f_x0 + it_hx[i]) < (unsigned)a.W;
    const int off = okx ? f_org + it_off[i] : (int)0xFFFFFFE0;
#ifndef DC_NO_FETCH
    pre[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rs, off, 0, 0));
    pre[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rs, off + 16, 0, 0));
#else
    pre[i][0] = pre[i][1] = f32x4{(float)off, 0.f, 0.f, 0.f};
#endif
  };
  auto convert_item = [&](int i) {
#ifdef DC_NO_STASH
    return;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = pre[i][j >> 2][j & 3];
#ifdef DC_H2
      const float xs = v * sc_x;
      const _Float16 hh = (_Float16)xs;
      pl[i][0][j] = hh;
      pl[i][1][j] = (_Float16)(xs - (float)hh);
#else
      const __bf16 hh = (__bf16)v;
      const float r1 = v - (float)hh;
      const __bf16 mm = (__bf16)r1;
      pl[i][0][j] = hh;
      pl[i][1][j] = mm;
#if DC_NPL == 3
      pl[i][2][j] = (__bf16)(r1 - (float)mm);
#endif
#endif
    }
    // pin the arithmetic to this K-step (LLVM would otherwise sink it to the stash, in front of the block-wide barrier)
#pragma unroll
    for (int p = 0; p < DC_NPL; ++p) {
      typedef int v4i __attribute__((ext_vector_type(4)));
      v4i t = __builtin_bit_cast(v4i, pl[i][p]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int te = t[e];
        asm volatile("" : "+v"(te));
        t[e] = te;
      }
      pl[i][p] = __builtin_bit_cast(bf16x8, t);
    }
  };
  auto stash = [&]() {
#ifdef DC_NO_STASH
    return;
#endif
#pragma unroll
    for (int i = 0; i < G::IPT; ++i) {
      const int px = item_px(i);
      const int dst = item_g8(i) * G::KGS + ((px / G::HW) * G::RW + px % G::HW) * 16;
      if (item_live(i)) {
        *(bf16x8*)(smem + dst) = pl[i][0];
        *(bf16x8*)(smem + dst + G::XPS) = pl[i][1];
        if (DC_NPL == 3) *(bf16x8*)(smem + dst + 2 * G::XPS) = pl[i][2];
      }
    }
  };

  // ---- weight stream: step q (0 .. NCHUNK NSTEP - 1, the same for every tile) of this co block -> LDS slot --------------------
  const unsigned char* wsrc = (const unsigned char*)a.wimg + (long long)cob * (G::NCHUNK * G::NSTEP) * G::WSTEP + lane * 16;
  auto wfill = [&](int q, int slot) {
#ifdef DC_NO_DMA
    return;
#endif
    unsigned char* dst = smem + G::XBYTES + slot * G::WSTEP;
    const unsigned char* src = wsrc + (long long)q * G::WSTEP;
#pragma unroll
    for (int i = 0; i < (G::WITEMS + G::NW - 1) / G::NW; ++i) {
      const int it = wave + i * G::NW;
      if (it < G::WITEMS) dc_glds16(src + it * 1024, dst + it * 1024);
    }
  };

  // ---- fragment addresses (tile independent) ----------------------------------------------------------------------------------
  auto prow = [&](int pt) { return ((wp * G::PB + pt) / G::TPW) * G::PH + i16 / G::PW; };
  auto pcol = [&](int pt) { return ((wp * G::PB + pt) % G::TPW) * G::PW + i16 % G::PW; };
  int xb[G::PB];
#pragma unroll
  for (int pt = 0; pt < G::PB; ++pt) xb[pt] = kg * G::KGS + (prow(pt) * G::RW + pcol(pt)) * 16;
  const int wb = kg * (G::NT * 16) + (wc * G::CB * 16 + i16) * 16;
  constexpr int XD = G::XPF ? 2 : 1;
  bf16x8 xf[XD][G::PB][DC_NPL], wf[G::CB][DC_NPL];
  auto xread1 = [&](const int s, const int set, const int pt) {   // pixel fragments of K-step s (current chunk image), pixel tile pt
    const Unit ua = unit_of(G::KC, s, 0), ub = unit_of(G::KC, s, 1);
    const int offa = ((ua.tap / 3) * G::RW + ua.tap % 3) * 16, offb = ((ub.tap / 3) * G::RW + ub.tap % 3) * 16;
    int o;
    if (s < 9) o = xb[pt] + offa + 2 * ua.win * G::KGS;
    else o = xb[pt] + (2 * ua.win - (kg & 2)) * G::KGS + (kg >> 1 ? offb : offa);   // group kg -> window 2, group kg & 1
#pragma unroll
    for (int p = 0; p < DC_NPL; ++p) xf[set][pt][p] = *(const bf16x8*)(smem + p * G::XPS + o);
  };
  auto xread = [&](const int s, const int set) {
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) xread1(s, set, pt);
  };
  auto wread1 = [&](const int slot, const int ct) {
    const unsigned char* wbuf = smem + G::XBYTES + slot * G::WSTEP;
#pragma unroll
    for (int p = 0; p < DC_NPL; ++p) wf[ct][p] = *(const bf16x8*)(wbuf + p * G::WPS + wb + ct * 256);
  };

  f32x4 acc[G::CB][G::PB];
  constexpr int TSTEPS = G::NCHUNK * G::NSTEP;               // K-steps per tile
  const int total = (t_end - t_begin) * TSTEPS;              // K-steps of this block
  target(t_begin, 0);
#pragma unroll
  for (int i = 0; i < G::IPT; ++i) fetch_item(i);
#pragma unroll
  for (int i = 0; i < G::IPT; ++i) convert_item(i);
  wfill(0, 0);
  if (total > 1) wfill(1 % TSTEPS, 1);
  int gs = 0;      // K-steps done: weights of step gs live in slot gs & 1
  int qn = 2 % TSTEPS;   // step-in-tile index of the next weight DMA (step gs + 2)

#pragma unroll 1
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const long long img0 = (long long)b * a.H * a.W;
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
      for (int pt = 0; pt < G::PB; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int chunk = 0; chunk < G::NCHUNK; ++chunk) {
      stash();                                   // (every wave's reads of the previous image are behind the last K-step's barrier)
      const bool last_chunk = chunk == G::NCHUNK - 1;
      const bool have_next = !(last_chunk && tile + 1 == t_end);
      if (have_next) target(last_chunk ? tile + 1 : tile, last_chunk ? 0 : chunk + 1);   // (no next: the current tile is fetched again, unused)
      DC_WAIT_VM(0);
      DC_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      if (gs == 0) {                             // (afterwards: rolled in under the previous K-step's MFMAs)
#pragma unroll
        for (int ct = 0; ct < G::CB; ++ct) wread1(0, ct);
        // the first K-step issues the LDS-DMA of step 2 into slot 0: every wave must hold its fragments of step 0 before that
        // (in the steady state the barrier that ends a K-step stands between a slot's last fragment read and its refill)
        DC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
      }
      xread(0, 0);
#pragma unroll
      for (int s = 0; s < G::NSTEP; ++s) {
        if (gs + 2 < total) {
          wfill(qn, gs & 1);
          qn = qn + 1 == TSTEPS ? 0 : qn + 1;
        }
        if (s < G::IPT) fetch_item(s);
#ifdef DC_SGB
        __builtin_amdgcn_sched_barrier(0);      // (region of the group barriers below: split arithmetic + fragment reads + MFMAs)
#endif
        if (s >= 2 && s - 2 < G::IPT) convert_item(s - 2);
        const int set = G::XPF ? (s & 1) : 0;
#ifndef DC_NO_XREAD
        if (G::XPF && s + 1 < G::NSTEP) xread(s + 1, (s + 1) & 1);
#endif
#pragma unroll
        for (int ct = 0; ct < G::CB; ++ct) {
#if !defined(DC_NOFENCE) && !defined(DC_SGB)
          __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef DC_SETPRIO
          __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
          for (int pt = 0; pt < G::PB; ++pt) {
            f32x4 c = acc[ct][pt];
#if DC_NPL == 3
            c = DC_MFMA(wf[ct][1], xf[set][pt][1], c, 0, 0, 0);   // m m
            c = DC_MFMA(wf[ct][0], xf[set][pt][2], c, 0, 0, 0);   // h l
            c = DC_MFMA(wf[ct][2], xf[set][pt][0], c, 0, 0, 0);   // l h
#endif
            c = DC_MFMA(wf[ct][0], xf[set][pt][1], c, 0, 0, 0);   // h m
            c = DC_MFMA(wf[ct][1], xf[set][pt][0], c, 0, 0, 0);   // m h
            c = DC_MFMA(wf[ct][0], xf[set][pt][0], c, 0, 0, 0);   // h h
            acc[ct][pt] = c;
#ifndef DC_NO_XREAD
            if (!G::XPF && ct == G::CB - 1 && s + 1 < G::NSTEP) {   // single fragment set: refilled behind its last use
#ifndef DC_SGB
              __builtin_amdgcn_sched_barrier(0);
#endif
              xread1(s + 1, 0, pt);
            }
#endif
          }
#ifdef DC_SETPRIO
          __builtin_amdgcn_s_setprio(0);
#endif
#if !defined(DC_NOFENCE) && !defined(DC_SGB)
          __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef DC_NO_WREAD
          wread1((gs + 1) & 1, ct);     // the next step's weight fragments roll in behind the last use of these registers
#endif
        }
#ifdef DC_SGB
        // issue order of the step's region: the pixel-fragment prefetch first, then every MFMA followed by two of the VALU operations
        // of the split arithmetic (they run in the shadow of the MFMA's own pipe time instead of as a block in front of the MFMAs),
        // the weight fragments of the next step behind each output-channel tile's last MFMA
        if (G::XPF && s + 1 < G::NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 3 * G::PB, 0);
#pragma unroll
        for (int ct = 0; ct < G::CB; ++ct) {
#pragma unroll
          for (int k = 0; k < 6 * G::PB; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
        ++gs;
        // the weights of step gs + 1 have landed (the two prefetch loads issued behind their DMA may stay in flight), every LDS
        // read of this step has returned
#ifndef DC_NO_SYNC
        if (s < G::IPT) DC_WAIT_VM(2);
        else DC_WAIT_VM(0);
        DC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
#endif
      }
    }

    // ---- epilogue: bias, store, BatchNorm partials ----------------------------------------------------------------------------
#ifdef DC_H2
    // back to the operands' scale, 2^-(e_x + e_w): one exact multiplication while that power of two is a normal float, else two steps
    if (ex_x + ex_w >= -120 && ex_x + ex_w <= 120) {
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
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) p_ok[pt] = y0 + prow(pt) < a.H && x0 + pcol(pt) < a.W;
    const int co0 = cob * G::NT + wc * G::CB * 16 + 4 * kg;   // + ct * 16 + r
    if constexpr (BQ) {
      dc_bnbwd_epilogue<G>(a, acc, p_ok, img0, y0, x0, co0, wc * G::CB * 16 + 4 * kg, wp, i16, tile, cob, smem, prow, pcol);
      continue;
    }
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bv = *(const f32x4*)(a.bias + co0 + ct * 16);
#pragma unroll
      for (int pt = 0; pt < G::PB; ++pt) {
        acc[ct][pt] += bv;
#ifdef DC_NO_STORE
        if (p_ok[pt] && acc[ct][pt][0] == 123.456f) {
#else
        if (p_ok[pt]) {
#endif
          float* dst = a.y + (img0 + (long long)(y0 + prow(pt)) * a.W + x0 + pcol(pt)) * a.ldy + co0 + ct * 16;
          f32x4 v = acc[ct][pt];
          if (a.accumulate) v += *(const f32x4*)dst;
          *(f32x4*)dst = v;
        }
      }
    }
    if (a.bn_part) {
      // (tile mean, sum(v - mean), sum((v - mean)^2)) per channel over the tile's valid pixels: two in-register passes; the 16 lanes
      // of a k-group hold 16 pixels of the same four channels, the WP waves of a channel group the other pixels
      float* scr = (float*)smem;    // [2 WP][NT]  (the K loop's last barrier is behind every LDS read of the image)
      const int nvalid = min(G::TH, a.H - y0) * min(G::TW, a.W - x0);
      const float inv = 1.f / (float)nvalid;
      const int cl = wc * G::CB * 16 + 4 * kg;   // block-local channel of (ct = 0, r = 0)
#pragma unroll
      for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = 0.f;
#pragma unroll
          for (int pt = 0; pt < G::PB; ++pt) v += p_ok[pt] ? acc[ct][pt][r] : 0.f;
          v = row16_sum(v);
          if (i16 == 0) scr[wp * G::NT + cl + ct * 16 + r] = v;
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
          d1 = row16_sum(d1);
          d2 = row16_sum(d2);
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
      __syncthreads();   // the scratch is read: the next tile's stash may overwrite it
    }
  }
}

// ---- wave-specialised variant: 4 compute waves + 4 helper waves per block -------------------------------------------------------
// In dconv3_b3_kernel every wave issues ~110 other instructions per K-step next to its 36 MFMAs (LDS-DMA, prefetch loads, the
// split arithmetic, address arithmetic): as many issue cycles as the MFMAs themselves.  Here the helper waves (4 .. 7) do all of
// that -- weight LDS-DMA, prefetch of the next halo tile, the fp32 -> 3 x bf16 split, the LDS image writes -- and the compute
// waves (0 .. 3) only read fragments and issue MFMAs (+ the epilogue).  Both roles run the same sequence of block barriers.
// 16 waves per CU (2 blocks): at most 128 registers per wave, which is why the two roles are separate code paths (no live range of
// one role overlaps the other's) and the pixel fragments are single-buffered.
template <class G, bool BQ = false>
__global__ __launch_bounds__(2 * G::NTHR, 4) void dconv3_b3_spec_kernel(const DcArgs a) {
  __shared__ __attribute__((aligned(256))) unsigned char smem[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kg = lane >> 4;
  const int cob = blockIdx.y;
  const int ntile = a.B * a.tiles_y * a.tiles_x;
#ifdef DC_H2
  const int ex_x = __builtin_amdgcn_readfirstlane(dc_exponent(cs_amax_read(a.x_rec)));   // prescale of the activation (its producer's amax record)
  const int ex_w = __builtin_amdgcn_readfirstlane(a.w_rec[1]);              // prescale of the weight image
  const float sc_x = __builtin_ldexpf(1.f, ex_x);                            // 2^e_x (|e| <= 100: a normal float; x * 2^e is exact)
#endif
  int t_begin, t_end;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    t_begin = (int)((long long)v * ntile / nb);
    t_end = (int)((long long)(v + 1) * ntile / nb);
  }
  if (t_begin >= t_end) return;
  constexpr int TSTEPS = G::NCHUNK * G::NSTEP;
  const int total = (t_end - t_begin) * TSTEPS;
  const bool bn = a.bn_part != nullptr;

  if (wave >= G::NW) {
    // =========================================================== helper role ===========================================================
    const int hw = wave - G::NW;
    auto item_px = [&](int i) { return ((hw + i * G::NW) / G::NGB) * 16 + i16; };
    auto item_g8 = [&](int i) { return ((hw + i * G::NW) % G::NGB) * 4 + kg; };
    auto item_live = [&](int i) { return hw + i * G::NW < G::NPG * G::NGB && item_px(i) < G::NPX && item_g8(i) < G::NKG; };
    int it_off[G::IPT], it_hx[G::IPT];
#pragma unroll
    for (int i = 0; i < G::IPT; ++i) {
      const int px = item_px(i);
      it_hx[i] = px % G::HW - 1;
      it_off[i] = (((px / G::HW - 1) * a.W + it_hx[i]) * a.ldx + (item_live(i) ? item_g8(i) * 8 : 0)) * 4;
    }
    f32x4 pre[G::IPT][2];
    bf16x8 pl[G::IPT][DC_NPL];
    const int img_bytes = ((a.H * a.W - 1) * a.ldx + G::KC) * 4;
    __amdgpu_buffer_rsrc_t f_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, img_bytes, 0x00020000);
    int f_x0 = 0, f_org = 0;
    auto target = [&](int tile, int chunk) {
      const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
      f_x0 = tx * G::TW;
      f_org = (ty * G::TH * a.W + f_x0) * a.ldx * 4;
      f_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)b * a.H * a.W * a.ldx + chunk * G::KC), (short)0, img_bytes, 0x00020000);
    };
    auto fetch_item = [&](int i) {
      const bool okx = (unsigned)(f_x0 + it_hx[i]) < (unsigned)a.W;
      const int off = okx ? f_org + it_off[i] : (int)0xFFFFFFE0;
      pre[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rs, off, 0, 0));
      pre[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f_rs, off + 16, 0, 0));
    };
    auto convert_item = [&](int i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = pre[i][j >> 2][j & 3];
#ifdef DC_H2
        const float xs = v * sc_x;
        const _Float16 hh = (_Float16)xs;
        pl[i][0][j] = hh;
        pl[i][1][j] = (_Float16)(xs - (float)hh);
#else
        const __bf16 hh = (__bf16)v;
        const float r1 = v - (float)hh;
        const __bf16 mm = (__bf16)r1;
        pl[i][0][j] = hh;
        pl[i][1][j] = mm;
#if DC_NPL == 3
        pl[i][2][j] = (__bf16)(r1 - (float)mm);
#endif
#endif
      }
#pragma unroll
      for (int p = 0; p < DC_NPL; ++p) {       // pin the arithmetic to this K-step
        typedef int v4i __attribute__((ext_vector_type(4)));
        v4i t = __builtin_bit_cast(v4i, pl[i][p]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int te = t[e];
          asm volatile("" : "+v"(te));
          t[e] = te;
        }
        pl[i][p] = __builtin_bit_cast(bf16x8, t);
      }
    };
    auto stash = [&]() {
#pragma unroll
      for (int i = 0; i < G::IPT; ++i) {
        const int px = item_px(i);
        const int dst = item_g8(i) * G::KGS + ((px / G::HW) * G::RW + px % G::HW) * 16;
        if (item_live(i)) {
          *(bf16x8*)(smem + dst) = pl[i][0];
          *(bf16x8*)(smem + dst + G::XPS) = pl[i][1];
          if (DC_NPL == 3) *(bf16x8*)(smem + dst + 2 * G::XPS) = pl[i][2];
        }
      }
    };
    const unsigned char* wsrc = (const unsigned char*)a.wimg + (long long)cob * (G::NCHUNK * G::NSTEP) * G::WSTEP + lane * 16;
    auto wfill = [&](int q, int slot) {
      unsigned char* dst = smem + G::XBYTES + slot * G::WSTEP;
      const unsigned char* src = wsrc + (long long)q * G::WSTEP;
#pragma unroll
      for (int i = 0; i < (G::WITEMS + G::NW - 1) / G::NW; ++i) {
        const int it = hw + i * G::NW;
        if (it < G::WITEMS) dc_glds16(src + it * 1024, dst + it * 1024);
      }
    };
    target(t_begin, 0);
#pragma unroll
    for (int i = 0; i < G::IPT; ++i) fetch_item(i);
#pragma unroll
    for (int i = 0; i < G::IPT; ++i) convert_item(i);
    wfill(0, 0);
    if (total > 1) wfill(1 % TSTEPS, 1);
    int gs = 0, qn = 2 % TSTEPS;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
#pragma unroll 1
      for (int chunk = 0; chunk < G::NCHUNK; ++chunk) {
        stash();
        const bool last_chunk = chunk == G::NCHUNK - 1;
        const bool have_next = !(last_chunk && tile + 1 == t_end);
        if (have_next) target(last_chunk ? tile + 1 : tile, last_chunk ? 0 : chunk + 1);
        DC_WAIT_VM(0);
        DC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        if (gs == 0) __builtin_amdgcn_s_barrier();          // (the compute waves hold their fragments of step 0: slot 0 may be refilled)
#pragma unroll
        for (int s = 0; s < G::NSTEP; ++s) {
          if (gs + 2 < total) {
            wfill(qn, gs & 1);
            qn = qn + 1 == TSTEPS ? 0 : qn + 1;
          }
          if (s < G::IPT) fetch_item(s);
          if (s >= 2 && s - 2 < G::IPT) convert_item(s - 2);
          ++gs;
          if (s < G::IPT) DC_WAIT_VM(2);
          else DC_WAIT_VM(0);
          __builtin_amdgcn_s_barrier();
        }
      }
      if constexpr (BQ) {         // the two barriers of dc_bnbwd_epilogue
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      } else if (bn) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }

  // ============================================================= compute role =============================================================
  const int wc = wave / G::WP, wp = wave % G::WP;
  auto prow = [&](int pt) { return ((wp * G::PB + pt) / G::TPW) * G::PH + i16 / G::PW; };
  auto pcol = [&](int pt) { return ((wp * G::PB + pt) % G::TPW) * G::PW + i16 % G::PW; };
  int xb[G::PB];
#pragma unroll
  for (int pt = 0; pt < G::PB; ++pt) xb[pt] = kg * G::KGS + (prow(pt) * G::RW + pcol(pt)) * 16;
  const int wb = kg * (G::NT * 16) + (wc * G::CB * 16 + i16) * 16;
  bf16x8 xf[G::PB][DC_NPL], wf[G::CB][DC_NPL];
  auto xread1 = [&](const int s, const int pt) {
    const Unit ua = unit_of(G::KC, s, 0), ub = unit_of(G::KC, s, 1);
    const int offa = ((ua.tap / 3) * G::RW + ua.tap % 3) * 16, offb = ((ub.tap / 3) * G::RW + ub.tap % 3) * 16;
    int o;
    if (s < 9) o = xb[pt] + offa + 2 * ua.win * G::KGS;
    else o = xb[pt] + (2 * ua.win - (kg & 2)) * G::KGS + (kg >> 1 ? offb : offa);
#pragma unroll
    for (int p = 0; p < DC_NPL; ++p) xf[pt][p] = *(const bf16x8*)(smem + p * G::XPS + o);
  };
  auto wread1 = [&](const int slot, const int ct) {
    const unsigned char* wbuf = smem + G::XBYTES + slot * G::WSTEP;
#pragma unroll
    for (int p = 0; p < DC_NPL; ++p) wf[ct][p] = *(const bf16x8*)(wbuf + p * G::WPS + wb + ct * 256);
  };
  f32x4 acc[G::CB][G::PB];
  int gs = 0;
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const long long img0 = (long long)b * a.H * a.W;
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
      for (int pt = 0; pt < G::PB; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int chunk = 0; chunk < G::NCHUNK; ++chunk) {
      __builtin_amdgcn_s_barrier();
      if (gs == 0) {
#pragma unroll
        for (int ct = 0; ct < G::CB; ++ct) wread1(0, ct);
        DC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int pt = 0; pt < G::PB; ++pt) xread1(0, pt);
#pragma unroll
      for (int s = 0; s < G::NSTEP; ++s) {
#pragma unroll
        for (int ct = 0; ct < G::CB; ++ct) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int pt = 0; pt < G::PB; ++pt) {
            f32x4 c = acc[ct][pt];
#if DC_NPL == 3
            c = DC_MFMA(wf[ct][1], xf[pt][1], c, 0, 0, 0);
            c = DC_MFMA(wf[ct][0], xf[pt][2], c, 0, 0, 0);
            c = DC_MFMA(wf[ct][2], xf[pt][0], c, 0, 0, 0);
#endif
            c = DC_MFMA(wf[ct][0], xf[pt][1], c, 0, 0, 0);
            c = DC_MFMA(wf[ct][1], xf[pt][0], c, 0, 0, 0);
            c = DC_MFMA(wf[ct][0], xf[pt][0], c, 0, 0, 0);
            acc[ct][pt] = c;
            if (ct == G::CB - 1 && s + 1 < G::NSTEP) {     // pixel fragments of the next step behind their last use
              __builtin_amdgcn_sched_barrier(0);
              xread1(s + 1, pt);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          wread1((gs + 1) & 1, ct);
        }
        ++gs;
        DC_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
      }
    }
    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
#ifdef DC_H2
    // back to the operands' scale, 2^-(e_x + e_w): one exact multiplication while that power of two is a normal float, else two steps
    if (ex_x + ex_w >= -120 && ex_x + ex_w <= 120) {
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
#pragma unroll
    for (int pt = 0; pt < G::PB; ++pt) p_ok[pt] = y0 + prow(pt) < a.H && x0 + pcol(pt) < a.W;
    const int co0 = cob * G::NT + wc * G::CB * 16 + 4 * kg;
    if constexpr (BQ) {
      dc_bnbwd_epilogue<G>(a, acc, p_ok, img0, y0, x0, co0, wc * G::CB * 16 + 4 * kg, wp, i16, tile, cob, smem, prow, pcol);
      continue;
    }
#pragma unroll
    for (int ct = 0; ct < G::CB; ++ct) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) bv = *(const f32x4*)(a.bias + co0 + ct * 16);
#pragma unroll
      for (int pt = 0; pt < G::PB; ++pt) {
        acc[ct][pt] += bv;
        if (p_ok[pt]) {
          float* dst = a.y + (img0 + (long long)(y0 + prow(pt)) * a.W + x0 + pcol(pt)) * a.ldy + co0 + ct * 16;
          f32x4 v = acc[ct][pt];
          if (a.accumulate) v += *(const f32x4*)dst;
          *(f32x4*)dst = v;
        }
      }
    }
    if (bn) {
      float* scr = (float*)smem;
      const int nvalid = min(G::TH, a.H - y0) * min(G::TW, a.W - x0);
      const float inv = 1.f / (float)nvalid;
      const int cl = wc * G::CB * 16 + 4 * kg;
#pragma unroll
      for (int ct = 0; ct < G::CB; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = 0.f;
#pragma unroll
          for (int pt = 0; pt < G::PB; ++pt) v += p_ok[pt] ? acc[ct][pt][r] : 0.f;
          v = row16_sum(v);
          if (i16 == 0) scr[wp * G::NT + cl + ct * 16 + r] = v;
        }
      DC_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
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
      DC_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
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
          d1 = row16_sum(d1);
          d2 = row16_sum(d2);
          if (i16 == 0) {
            scr[wp * G::NT + cl + ct * 16 + r] = d1;
            scr[(G::WP + wp) * G::NT + cl + ct * 16 + r] = d2;
          }
        }
      DC_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
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
      DC_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
    }
  }
}

// OHWI fp32 weights [C][3][3][C] -> the kernel's weight image, split into three bf16 planes.
//   forward:       A[co][k = (tap, c)]  = w[co][tap][c]
//   backward-data: A[ci][k = (tap, o)]  = w[o][8 - tap][ci]      (dx = conv of dy with the transposed, tap-mirrored bank)
__device__ __forceinline__ void dconv3_prep_body(const float* __restrict__ w, int C, int KC, int NT, int dgrad, u16* __restrict__ img,
                                                 long long first, long long stride, int e) {
  const int nstep = steps_of(KC), nchunk = C / KC;
  const long long total = (long long)(C / NT) * nchunk * nstep * NT * 32;
  for (long long i = first; i < total; i += stride) {
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
#ifdef DC_H2
    const long long base = (((long long)(cob * nchunk + chunk) * nstep + step) * 2) * (32LL * NT) + ((long long)g * NT + co) * 8 + j;
    const float xs = __builtin_ldexpf(v, e);
    const _Float16 hh = (_Float16)xs;
    img[base] = __builtin_bit_cast(u16, hh);
    img[base + 32LL * NT] = __builtin_bit_cast(u16, (_Float16)(xs - (float)hh));
#else
    (void)e;
    const __bf16 hh = (__bf16)v;
    const float r1 = v - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const __bf16 ll = (__bf16)(r1 - (float)mm);
    const long long base = (((long long)(cob * nchunk + chunk) * nstep + step) * 3) * (32LL * NT) + ((long long)g * NT + co) * 8 + j;
    img[base] = __builtin_bit_cast(u16, hh);
    img[base + 32LL * NT] = __builtin_bit_cast(u16, mm);
    img[base + 64LL * NT] = __builtin_bit_cast(u16, ll);
#endif
  }
}

// all layers of a network in one launch: entry e = {weight offset (floats) in the flat parameter buffer, C, direction, image offset
// (bytes)}, blockIdx.y = entry.  DC_H2: recs[2 * entry] = bits of the layer's max |w| (dconv3_h2_amax_batch_kernel), the exponent
// derived from it goes to recs[2 * entry + 1] (read by the convolution's epilogue)
struct DcPrepEntry { long long w_off; long long img_off; int C, KC, NT, dgrad; };
#ifdef DC_H2
__global__ __launch_bounds__(256) void dconv3_h2_amax_batch_kernel(const float* __restrict__ flat, const DcPrepEntry* __restrict__ ent,
                                                                   unsigned* __restrict__ recs) {
  const DcPrepEntry e = ent[blockIdx.y];
  const float* w = flat + e.w_off;
  const long long n = (long long)e.C * 9 * e.C;
  unsigned m = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m = max(m, __float_as_uint(w[i]) & 0x7FFFFFFFu);
  cs_amax_commit1(m, recs + 2 * blockIdx.y);
}
__global__ __launch_bounds__(256) void dconv3_prep_batch_kernel(const float* __restrict__ flat, const DcPrepEntry* __restrict__ ent,
                                                                unsigned char* __restrict__ img_base, unsigned* __restrict__ recs) {
  const DcPrepEntry e = ent[blockIdx.y];
  const int ex = dc_exponent(recs[2 * blockIdx.y]);
  if (blockIdx.x == 0 && threadIdx.x == 0) ((int*)recs)[2 * blockIdx.y + 1] = ex;
  dconv3_prep_body(flat + e.w_off, e.C, e.KC, e.NT, e.dgrad, (u16*)(img_base + e.img_off), blockIdx.x * (long long)blockDim.x + threadIdx.x,
                   (long long)gridDim.x * blockDim.x, ex);
}
#else
__global__ __launch_bounds__(256) void dconv3_prep_kernel(const float* __restrict__ w, int C, int KC, int NT, int dgrad, u16* __restrict__ img) {
  dconv3_prep_body(w, C, KC, NT, dgrad, img, blockIdx.x * (long long)blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x, 0);
}
__global__ __launch_bounds__(256) void dconv3_prep_batch_kernel(const float* __restrict__ flat, const DcPrepEntry* __restrict__ ent,
                                                                unsigned char* __restrict__ img_base) {
  const DcPrepEntry e = ent[blockIdx.y];
  dconv3_prep_body(flat + e.w_off, e.C, e.KC, e.NT, e.dgrad, (u16*)(img_base + e.img_off), blockIdx.x * (long long)blockDim.x + threadIdx.x,
                   (long long)gridDim.x * blockDim.x, 0);
}
#endif

//                     C  KC  NT WC WP PB TPH TPW PH PW  XPF
#ifndef DC_SPEC48      // (A/B hook: wave specialisation for 48 / 64 channels.  bf16x3 build: 65 -> 70 us; f16x2 build: 53 / 52 -> 54 / 50 us at 48
#define DC_SPEC48 false  //  channels, 72 -> 84 us at 64: not adopted in either)
#endif
using Cfg48 = DcCfg<48, 48, 48, 1, 4, 2, 8, 1, 1, 16, true, DC_SPEC48>;    // tile  8 x 16, wave = 48 co x 32 px
using Cfg64 = DcCfg<64, 32, 64, 1, 4, 2, 8, 1, 1, 16, true, DC_SPEC48>;    // tile  8 x 16, wave = 64 co x 32 px (the stage-1 bottlenecks' 3x3)
using Cfg96 = DcCfg<96, 32, 96, 2, 2, 4, 4, 2, 1, 16, false>;    // tile  4 x 32, wave = 48 co x 64 px (A/B alternative)

using Cfg96s = DcCfg<96, 32, 96, 2, 2, 2, 4, 1, 1, 16, true, true>;   // tile 4 x 16, wave = 48 co x 32 px, specialised waves (the default for 96)
using Cfg192 = DcCfg<192, 32, 96, 2, 2, 2, 2, 2, 1, 16, true, true>;  // tile 2 x 32, wave = 48 co x 32 px, two co blocks; specialised waves
using Cfg384 = DcCfg<384, 32, 96, 2, 2, 2, 2, 2, 1, 16, true, true>;  // the same tile, four co blocks

struct DcPlan { int kind, KC, NT, TH, TW; };

}  // namespace

// Tuning hooks (catseg_debug_set_dconv3_*).  This source compiles twice (see the head of the file): the variables have external linkage and
// live in the bf16x3 object, the f16x2 object refers to the same ones -- a hook set once acts on both builds.
#ifndef DC_H2
int catseg_g_dc_blocks = 512;   // persistent blocks per launch: two per CU
int catseg_g_dc_alt96 = 0;      // A/B: 1 = the 4 x 32-pixel configuration with uniform waves for 96 channels (default: 4 x 16, specialised waves)
int catseg_g_dc_spec = -1;      // -1: per configuration (Cfg::SPEC), 0 / 1: force the uniform / the wave-specialised kernel
#else
extern int catseg_g_dc_blocks, catseg_g_dc_alt96, catseg_g_dc_spec;
#endif
#define g_dc_blocks catseg_g_dc_blocks
#define g_dc_alt96 catseg_g_dc_alt96
#define g_dc_spec catseg_g_dc_spec

namespace {

DcPlan dc_plan(int C) {
  if (C == 48) return {1, Cfg48::KC, Cfg48::NT, Cfg48::TH, Cfg48::TW};
  if (C == 64) return {5, Cfg64::KC, Cfg64::NT, Cfg64::TH, Cfg64::TW};
  if (C == 96 && g_dc_alt96) return {2, Cfg96::KC, Cfg96::NT, Cfg96::TH, Cfg96::TW};
  if (C == 96) return {6, Cfg96s::KC, Cfg96s::NT, Cfg96s::TH, Cfg96s::TW};     // 65 / 62 us against 71 / 68 us for the 4 x 32 tile with uniform waves
  if (C == 192) return {3, Cfg192::KC, Cfg192::NT, Cfg192::TH, Cfg192::TW};
  if (C == 384) return {4, Cfg384::KC, Cfg384::NT, Cfg384::TH, Cfg384::TW};
  return {0, 0, 0, 0, 0};
}

template <class G>
int dc_launch(const DcArgs& a, int C, hipStream_t st) {
  const int ntile = a.B * a.tiles_y * a.tiles_x;
  // persistent blocks pay when a block gets >= 3 tiles (the next halo tile is prefetched under the current one: 2040 tiles of the
  // 48-channel layer, 56.7 us with 512 blocks against 60.9 us with one block per tile); with fewer tiles per slot the uneven
  // split costs more (544 tiles of the 96-channel layer: 79 us with 512 blocks, 65 us with 544)
  const int nb = ntile > 3 * g_dc_blocks ? g_dc_blocks : ntile;
  // specialised waves: 192 / 384 channels 82 -> 73 us, 90 -> 79 us; 48 channels 65 -> 70 us (its helper waves carry six staging items per
  // 14-step tile and the role needs 9 spilled registers), 96 channels: the 48 x 64 wave tile does not fit 128 registers
  const bool spec = g_dc_spec < 0 ? G::SPEC : g_dc_spec != 0;
  if (a.bq_part) {
    if (spec) hipLaunchKernelGGL((dconv3_b3_spec_kernel<G, true>), dim3(nb, C / G::NT), dim3(2 * G::NTHR), 0, st, a);
    else hipLaunchKernelGGL((dconv3_b3_kernel<G, true>), dim3(nb, C / G::NT), dim3(G::NTHR), 0, st, a);
  } else if (spec) hipLaunchKernelGGL((dconv3_b3_spec_kernel<G>), dim3(nb, C / G::NT), dim3(2 * G::NTHR), 0, st, a);
  else hipLaunchKernelGGL((dconv3_b3_kernel<G>), dim3(nb, C / G::NT), dim3(G::NTHR), 0, st, a);
  return 0;
}

}  // namespace

namespace {
int dc_run(int B, int H, int W, int C, const float* x, int ldx, const void* wimg, const float* bias, float* y, int ldy, int accumulate,
           float* bn_part, size_t bn_part_floats, int* bn_counts, const float* bq_y, int bq_ldy, const float* bq_stats,
           const float* bq_gamma, const float* bq_beta, float* bq_part, size_t bq_part_floats, const void* x_rec, const void* w_rec,
           catseg_stream_t stream);
}

#ifndef DC_H2
extern "C" int catseg_debug_set_dconv3_blocks(int blocks) {
  g_dc_blocks = blocks > 0 ? blocks : 512;
  return CATSEG_OK;
}

extern "C" int catseg_debug_set_dconv3_alt96(int on) {
  g_dc_alt96 = on ? 1 : 0;
  return CATSEG_OK;
}

extern "C" int catseg_debug_set_dconv3_spec(int on) {
  g_dc_spec = on < 0 ? -1 : (on ? 1 : 0);
  return CATSEG_OK;
}

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

// entries: DEVICE array of n records {int64 weight offset (floats, relative to flat), int64 image offset (bytes, relative to
// wimg_base), int32 C, int32 KC, int32 NT, int32 backward_data}; KC / NT as catseg_dconv3_layout reports them for C
extern "C" int catseg_dconv3_prep_batch(const float* flat, int n, const void* entries, void* wimg_base, catseg_stream_t stream) {
  CS_REQUIRE(flat && entries && wimg_base && n > 0, "dconv3 prep batch: bad args");
  hipLaunchKernelGGL(dconv3_prep_batch_kernel, dim3(48, n), dim3(256), 0, (hipStream_t)stream, flat, (const DcPrepEntry*)entries,
                     (unsigned char*)wimg_base);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

extern "C" int catseg_dconv3_layout(int C, int* kc, int* nt) {
  const DcPlan p = dc_plan(C);
  if (!p.kind) return 0;
  if (kc) *kc = p.KC;
  if (nt) *nt = p.NT;
  return 1;
}


extern "C" int catseg_dconv3(int B, int H, int W, int C, const float* x, int ldx, const void* wimg, const float* bias, float* y, int ldy,
                             int accumulate, float* bn_part, size_t bn_part_floats, int* bn_counts, catseg_stream_t stream) {
  return dc_run(B, H, W, C, x, ldx, wimg, bias, y, ldy, accumulate, bn_part, bn_part_floats, bn_counts, nullptr, 0, nullptr, nullptr, nullptr,
                nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int catseg_dconv3_bnbwd(int B, int H, int W, int C, const float* dy, int lddy, const void* wimg_bwd, float* g, int ldg,
                                   const float* q, int ldq, const float* stats, const float* gamma, const float* beta, float* part,
                                   size_t part_floats, catseg_stream_t stream) {
  CS_REQUIRE(q && stats && gamma && beta && part, "dconv3 bnbwd: bad args");
  CS_REQUIRE(ldq >= C && ldq % 4 == 0 && cs_aligned16(q) && cs_aligned16(stats) && cs_aligned16(gamma) && cs_aligned16(beta) &&
                 cs_aligned16(part), "dconv3 bnbwd: alignment / row strides");
  CS_REQUIRE((long long)H * W * (long long)ldq < (1LL << 31), "dconv3 bnbwd: image too large for 32-bit offsets");
  return dc_run(B, H, W, C, dy, lddy, wimg_bwd, nullptr, g, ldg, 0, nullptr, 0, nullptr, q, ldq, stats, gamma, beta, part, part_floats, nullptr, nullptr, stream);
}

#else   // ---------------------------------------------------------------- DC_H2: the two-plane fp16 build (dconv3_f16x2.hip)
extern "C" size_t catseg_dconv3_f16x2_wimg_bytes(int C) {
  const DcPlan p = dc_plan(C);
  if (!p.kind) return 0;
  return (size_t)(C / p.NT) * (C / p.KC) * steps_of(p.KC) * 128 * p.NT;
}

// as catseg_dconv3_prep_batch; records = n x {uint32 bits of max|w|, int32 exponent} (DEVICE, 8 bytes per entry): zeroed, filled by
// the amax launch, completed by the image launch.  Three stream operations for all layers of a network.
extern "C" int catseg_dconv3_f16x2_prep_batch(const float* flat, int n, const void* entries, void* wimg_base, void* records,
                                              catseg_stream_t stream) {
  CS_REQUIRE(flat && entries && wimg_base && records && n > 0, "dconv3 f16x2 prep batch: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(records, 0, (size_t)n * 8, st) != hipSuccess) { catseg_set_error("dconv3 f16x2 prep: memset failed"); return CATSEG_EHIP; }
  hipLaunchKernelGGL(dconv3_h2_amax_batch_kernel, dim3(16, n), dim3(256), 0, st, flat, (const DcPrepEntry*)entries, (unsigned*)records);
  hipLaunchKernelGGL(dconv3_prep_batch_kernel, dim3(48, n), dim3(256), 0, st, flat, (const DcPrepEntry*)entries, (unsigned char*)wimg_base,
                     (unsigned*)records);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}

// catseg_dconv3 on two fp16 planes: x_record = the activation's amax record (16 words 128 bytes apart, their max = bits of max|x| over
// the WHOLE tensor, as its producer left it: catseg_bn_apply_amax, catseg_add_n_act_amax, catseg_bn_backward(_pre)_amax; a larger value is safe, a
// smaller one overflows fp16), w_record = the weight image's record from catseg_dconv3_f16x2_prep_batch
extern "C" int catseg_dconv3_f16x2(int B, int H, int W, int C, const float* x, int ldx, const void* x_record, const void* wimg,
                                   const void* w_record, const float* bias, float* y, int ldy, int accumulate, float* bn_part,
                                   size_t bn_part_floats, int* bn_counts, catseg_stream_t stream) {
  CS_REQUIRE(x_record && w_record, "dconv3 f16x2: amax records missing");
  return dc_run(B, H, W, C, x, ldx, wimg, bias, y, ldy, accumulate, bn_part, bn_part_floats, bn_counts, nullptr, 0, nullptr, nullptr, nullptr,
                nullptr, 0, x_record, w_record, stream);
}

extern "C" int catseg_dconv3_bnbwd_f16x2(int B, int H, int W, int C, const float* dy, int lddy, const void* dy_record, const void* wimg_bwd,
                                         const void* w_record, float* g, int ldg, const float* q, int ldq, const float* stats,
                                         const float* gamma, const float* beta, float* part, size_t part_floats, catseg_stream_t stream) {
  CS_REQUIRE(q && stats && gamma && beta && part && dy_record && w_record, "dconv3 bnbwd f16x2: bad args");
  CS_REQUIRE(ldq >= C && ldq % 4 == 0 && cs_aligned16(q) && cs_aligned16(stats) && cs_aligned16(gamma) && cs_aligned16(beta) &&
                 cs_aligned16(part), "dconv3 bnbwd f16x2: alignment / row strides");
  CS_REQUIRE((long long)H * W * (long long)ldq < (1LL << 31), "dconv3 bnbwd f16x2: image too large for 32-bit offsets");
  return dc_run(B, H, W, C, dy, lddy, wimg_bwd, nullptr, g, ldg, 0, nullptr, 0, nullptr, q, ldq, stats, gamma, beta, part, part_floats, dy_record,
                w_record, stream);
}
#endif

namespace {
int dc_run(int B, int H, int W, int C, const float* x, int ldx, const void* wimg, const float* bias, float* y, int ldy, int accumulate,
           float* bn_part, size_t bn_part_floats, int* bn_counts, const float* bq_y, int bq_ldy, const float* bq_stats,
           const float* bq_gamma, const float* bq_beta, float* bq_part, size_t bq_part_floats, const void* x_rec, const void* w_rec,
           catseg_stream_t stream) {
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
  a.bq_y = bq_y; a.bq_ldy = bq_ldy; a.bq_stats = bq_stats; a.bq_gamma = bq_gamma; a.bq_beta = bq_beta; a.bq_part = bq_part;
  a.x_rec = (const unsigned*)x_rec; a.w_rec = (const int*)w_rec;
  const long long ntile = (long long)B * a.tiles_y * a.tiles_x;
  if (bn_part) CS_REQUIRE(bn_counts && bn_part_floats >= (size_t)ntile * 3 * C, "dconv3: BatchNorm partial buffer too small");
  if (bq_part) CS_REQUIRE(bq_part_floats >= (size_t)ntile * 2 * C, "dconv3 bnbwd: partial buffer too small");
  if (p.kind == 1) dc_launch<Cfg48>(a, C, (hipStream_t)stream);
  else if (p.kind == 2) dc_launch<Cfg96>(a, C, (hipStream_t)stream);
  else if (p.kind == 3) dc_launch<Cfg192>(a, C, (hipStream_t)stream);
  else if (p.kind == 5) dc_launch<Cfg64>(a, C, (hipStream_t)stream);
  else if (p.kind == 6) dc_launch<Cfg96s>(a, C, (hipStream_t)stream);
  else dc_launch<Cfg384>(a, C, (hipStream_t)stream);
  CS_LAUNCH_CHECK();
  return CATSEG_OK;
}
}  // namespace
