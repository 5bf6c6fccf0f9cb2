// Host-side logic of libcatseg_hip.so under AddressSanitizer + UBSan (the reference has no sanitizer; "the build adds ... an ASan
// build of the C++ host code").  GPU code cannot be sanitised on this pool, and needs no GPU here: everything called below either answers on
// the host (size / plan / capability queries) or must REJECT its arguments before the first HIP call.  Built and run by
// tests/test_host_asan_cpu.py against csrc/Makefile's `asan` target; test infrastructure, not shipped.
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "catseg.h"
#include "catseg_debug.h"

static int checks = 0, failed = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    ++checks;                                                               \
    if (!(cond)) { ++failed; std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)

static catseg_conv_desc desc(int B, int H, int W, int Cin, int Cout, int k, int s, int p, int d = 1) {
  catseg_conv_desc c;
  std::memset(&c, 0, sizeof c);
  c.B = B; c.H = H; c.W = W; c.Cin = Cin; c.Cout = Cout;
  c.kh = c.kw = k; c.stride = s; c.pad = p; c.dil = d;
  c.Ho = (H + 2 * p - d * (k - 1) - 1) / s + 1;
  c.Wo = (W + 2 * p - d * (k - 1) - 1) / s + 1;
  c.ldx = Cin; c.ldy = (Cout + 3) / 4 * 4;
  return c;
}

int main() {
  EXPECT(catseg_version() >= 1);
  alignas(16) static float buf[64];
  float* p16 = buf;

  // ---- size / capability queries over the shapes of the bench model and a few hostile ones
  const int widths[] = {0, 1, 3, 4, 8, 25, 32, 48, 64, 96, 192, 384, 720, 2048, 1 << 20};
  for (int C : widths) {
    const int sup = catseg_dconv3_supported(C), supw = catseg_dwgrad3_supported(C);
    const int supp = catseg_dconv3_pl_supported(C), supwp = catseg_dwgrad3_pl_supported(C);
    EXPECT((sup == 0 || sup == 1) && (supw == 0 || supw == 1) && (supp == 0 || supp == 1) && (supwp == 0 || supwp == 1));
    if (sup) {
      EXPECT(catseg_dconv3_wimg_bytes(C) >= (size_t)9 * C * C * 2);
      int th = -1, tw = -1;
      const int nt = catseg_dconv3_tiles(C, 8, 136, 240, &th, &tw);
      EXPECT(nt > 0 && th > 0 && tw > 0 && (long long)nt * th * tw >= 8LL * 136 * 240);
      EXPECT(catseg_dconv3_tiles(C, 8, 136, 240, nullptr, nullptr) == nt);
      EXPECT(catseg_dconv3_tiles(C, 1, 1, 1, &th, &tw) >= 1);
    }
    if (supw) EXPECT(catseg_dwgrad3_workspace(8, 136, 240, C) > 0);
    if (supp) EXPECT(catseg_dconv3_pl_rows(C, 8, 136, 240) > 0 && catseg_dconv3_pl_rows(C, 1, 3, 5) > 0);
    if (supwp) EXPECT(catseg_dwgrad3_pl_workspace(8, 68, 120, C) > 0);
    if (C > 0 && C % 8 == 0 && C <= 4096) EXPECT(catseg_planes_bytes(1000, C) == (size_t)1000 * C * 4);
  }
  EXPECT(catseg_split3_elems(10, 13) == 10u * 16u);
  EXPECT(catseg_split3_blocked_elems(10, 17) == 3u * 32u * 10u);
  EXPECT(catseg_split2h_blocked_elems(7, 48) == 2u * 48u * 7u && catseg_split2h_planar_elems(7, 50) == 2u * 7u * 56u);
  EXPECT(catseg_lovasz_workspace(4177920, 25) > (size_t)4177920 * 25 * 16);
  EXPECT(catseg_lovasz_workspace(1, 1) > 0 && catseg_ce_workspace(1) > 0 && catseg_ohem_workspace(1) > 0);
  EXPECT(catseg_ce_workspace(4177920) > 0 && catseg_ohem_workspace(4177920) > 0);
  EXPECT(catseg_bn_workspace(261120, 48) > 0 && catseg_bn_workspace(1, 4) > 0 && catseg_bn_workspace(8LL * 544 * 960, 2048) > 0);
  EXPECT(catseg_softmax_spatial_workspace(8, 32640) > 0);

  // ---- tile planner over every convolution family of the three training configurations (+ degenerate extents)
  struct Shape { int B, H, W, Cin, Cout, k, s, p, d; };
  const Shape shapes[] = {
      {8, 544, 960, 4, 64, 3, 2, 1, 1},    {8, 272, 480, 64, 64, 3, 2, 1, 1},   {8, 136, 240, 64, 256, 1, 1, 0, 1},
      {8, 136, 240, 48, 48, 3, 1, 1, 1},   {8, 136, 240, 48, 96, 3, 2, 1, 1},   {8, 68, 120, 96, 192, 3, 2, 1, 1},
      {8, 17, 30, 384, 48, 1, 1, 0, 1},    {8, 136, 240, 720, 512, 3, 1, 1, 1}, {8, 136, 240, 1024, 512, 1, 1, 0, 1},
      {8, 136, 240, 512, 25, 1, 1, 0, 1},  {8, 68, 120, 2048, 256, 3, 1, 12, 12}, {8, 68, 120, 2048, 256, 3, 1, 36, 36},
      {1, 1, 1, 4, 1, 1, 1, 0, 1},         {1, 3, 5, 8, 7, 3, 1, 1, 1},         {2, 16, 24, 32, 32, 16, 8, 4, 1},
      {8, 25, 1, 512, 256, 1, 1, 0, 1},    {4, 272, 480, 2048, 512, 3, 1, 1, 1}};
  for (const Shape& q : shapes) {
    catseg_conv_desc d = desc(q.B, q.H, q.W, q.Cin, q.Cout, q.k, q.s, q.p, q.d);
    for (int op = 0; op < 3; ++op) {
      int out[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
      const int rc = catseg_debug_plan_conv(&d, op, out);
      if (rc != CATSEG_OK) std::printf("plan_conv(%d x %d x %d x %d -> %d, k%d s%d, op %d): %s\n", q.B, q.H, q.W, q.Cin, q.Cout, q.k, q.s, op, catseg_last_error());
      EXPECT(rc == CATSEG_OK);
      EXPECT(out[0] >= 0);
    }
    EXPECT(catseg_conv2d_bwd_weight_workspace(&d) > 0);
    if (q.Cin % 8 == 0) {       // (0 = this layer is not one the split-precision backward-weight kernels take)
      catseg_conv_desc e = d;
      e.ldx = q.Cin; e.ldy = (q.Cout + 7) / 8 * 8;
      (void)catseg_conv2d_bwd_weight_bf16x3_workspace(&e);
      (void)catseg_conv2d_bwd_weight_f16x2_workspace(&e);
    }
  }
  {
    catseg_conv_desc d = desc(1, 8, 8, 8, 8, 3, 1, 1);
    int out[8];
    EXPECT(catseg_debug_plan_conv(&d, 3, out) != CATSEG_OK && catseg_debug_plan_conv(&d, -1, out) != CATSEG_OK);
    EXPECT(catseg_debug_plan_conv(&d, 0, nullptr) != CATSEG_OK);
  }

  // ---- argument validation: every call below must fail ON THE HOST, with a message, before any HIP call (there is no GPU here)
  {
    catseg_conv_desc d = desc(1, 8, 8, 6, 16, 1, 1, 0);        // Cin not a multiple of 4
    EXPECT(catseg_conv2d_fwd(&d, p16, p16, nullptr, p16, 0, nullptr) != CATSEG_OK);
    EXPECT(std::strstr(catseg_last_error(), "multiple of 4") != nullptr);
    d = desc(1, 8, 8, 8, 16, 3, 1, 1);
    d.Ho = 9;                                                   // inconsistent geometry
    EXPECT(catseg_conv2d_fwd(&d, p16, p16, nullptr, p16, 0, nullptr) != CATSEG_OK);
    EXPECT(catseg_conv2d_bwd_data(&d, p16, p16, p16, 0, nullptr) != CATSEG_OK);
    d = desc(1, 8, 8, 8, 16, 3, 1, 1);
    d.ldx = 6;                                                  // row stride below the channel count
    EXPECT(catseg_conv2d_fwd(&d, p16, p16, nullptr, p16, 0, nullptr) != CATSEG_OK);
    d = desc(1, 8, 8, 8, 16, 3, 1, 1);
    EXPECT(catseg_conv2d_fwd(&d, p16 + 1, p16, nullptr, p16, 0, nullptr) != CATSEG_OK);      // misaligned pointer
    EXPECT(catseg_conv2d_fwd(&d, p16, p16, nullptr, p16, 64, nullptr) != CATSEG_OK);         // zero_to beyond the row
    d = desc(INT_MAX / 2, 8, 8, 8, 16, 3, 1, 1);                                            // extents whose products overflow 32 bits
    (void)catseg_conv2d_bwd_weight_workspace(&d);
    int out[8];
    (void)catseg_debug_plan_conv(&d, 0, out);
    d = desc(1, 8, 8, 8, 16, 3, 1, 1);
    d.stride = 0;                                                                           // stride 0: no division by it on the host
    EXPECT(catseg_conv2d_fwd(&d, p16, p16, nullptr, p16, 0, nullptr) != CATSEG_OK);
    d = desc(1, 8, 8, 8, 16, 3, 1, 1);
    d.groups = 3;                                                                           // channels not divisible by the groups
    EXPECT(catseg_conv2d_fwd(&d, p16, p16, nullptr, p16, 0, nullptr) != CATSEG_OK);
  }
  EXPECT(catseg_lovasz_softmax(p16, (const int64_t*)p16, 100, 200, 1.0f, p16, nullptr, 0, p16, (size_t)1 << 30, nullptr) != CATSEG_OK);   // K > 64
  EXPECT(catseg_lovasz_softmax(p16, (const int64_t*)p16, 100, 8, 1.0f, p16, nullptr, 0, p16, 16, nullptr) != CATSEG_OK);                  // workspace too small
  EXPECT(catseg_maxpool2x2_fwd(p16, 6, p16, 8, (uint8_t*)p16, 1, 4, 4, 8, nullptr) != CATSEG_OK);        // row stride not 16-byte granular
  EXPECT(catseg_maxpool2x2_fwd(p16, 8, p16, 8, (uint8_t*)p16, 1, 1, 4, 8, nullptr) != CATSEG_OK);        // no output row
  EXPECT(catseg_maxpool2x2_bwd(p16, 8, nullptr, p16, 8, 1, 4, 4, 8, nullptr) != CATSEG_OK);
  EXPECT(catseg_bias_rows(p16, p16, 4, 10, 8, nullptr) != CATSEG_OK);                                    // ld < C
  EXPECT(catseg_bias_grad(p16, 8, 10, 8, p16, p16, 16, nullptr) != CATSEG_OK);                           // workspace too small
  EXPECT(catseg_axpy2d(p16, 6, p16, 8, 4, 6, 1.0f, 0, nullptr) != CATSEG_OK);
  EXPECT(catseg_dwgrad3(8, 16, 16, 50, p16, 50, p16, 50, p16, p16, 1 << 20, nullptr) != CATSEG_OK);       // unsupported width
  EXPECT(catseg_planes_from_f32(p16, 6, 10, 6, p16, p16, 1, nullptr) != CATSEG_OK);                      // C not a multiple of 8
  EXPECT(catseg_dconv3_pl(1, 8, 8, 50, p16, p16, p16, p16, nullptr, p16, 50, 0, nullptr, 0, nullptr, nullptr, nullptr) != CATSEG_OK);
  EXPECT(std::strlen(catseg_last_error()) > 0);

  // ---- debug knobs: out-of-range values are refused or clamped, defaults restored
  (void)catseg_debug_set_tile(99, 99);
  (void)catseg_debug_set_tile(0, 0);
  (void)catseg_debug_set_splits(-5);
  (void)catseg_debug_set_splits(0);
  (void)catseg_debug_set_dconv3_blocks(-1);
  (void)catseg_debug_set_dconv3_blocks(0);
  (void)catseg_debug_set_dwgrad3_blocks(1 << 30);
  (void)catseg_debug_set_dwgrad3_blocks(0);
  (void)catseg_debug_set_dconv3_pl_slots(-3);
  (void)catseg_debug_set_dconv3_pl_slots(512);
  (void)catseg_debug_set_b3_tile(7);
  (void)catseg_debug_set_b3_tile(0);
  for (const Shape& q : shapes) {     // the planner again after the knobs went through their extremes
    catseg_conv_desc d = desc(q.B, q.H, q.W, q.Cin, q.Cout, q.k, q.s, q.p, q.d);
    int out[8];
    EXPECT(catseg_debug_plan_conv(&d, 2, out) == CATSEG_OK);
  }
  std::printf("%s %d checks, %d failed\n", failed ? "FAILED" : "ok", checks, failed);
  return failed ? 1 : 0;
}
