/*
 * catseg_debug.h — tuning / measurement hooks of libcatseg_hip.so.
 *
 * NOT part of the drop-in boundary (include/catseg.h): these knobs are process-global, not
 * thread-safe, and exist for tools/ (tile sweeps), bench.py (worst-case Lovasz timing) and the
 * tests that pin the planner's bench-size tile choices on small inputs.  Nothing in the package's
 * model / loss / manager code calls them.
 */
#ifndef CATSEG_DEBUG_H
#define CATSEG_DEBUG_H

#include "catseg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tuning hook: force the igemm block tile to (64*mi) x (64*ni); mi = 0 restores the heuristic */
int catseg_debug_set_tile(int mi, int ni);
/* tuning hook: force the backward-weight split count (0 restores the planner) */
int catseg_debug_set_splits(int splits);
/* measurement hook: 0 = stride-2 backward-data as one launch per input-pixel parity class (round 1), 1 (default) = all classes in
 * one launch (igemm_f32_multi_kernel) */
int catseg_debug_set_strided_multi(int on);
/* measurement hook: 0 = sort every pixel of every present class in catseg_lovasz_softmax (the data-independent worst case);
 * 1 (default) = sort only the elements that can precede the last foreground pixel (bit-identical result) */
int catseg_debug_set_lovasz_prune(int on);
/* measurement hook: 0 = 3x3 48->48 / 96->96 backward-weight through the implicit GEMM instead of the direct kernel,
 * 1 = direct kernel (default); a value > 1 additionally sets the direct kernel's target block count (default 512) */
int catseg_debug_set_wgrad_direct(int on);

/* tuning hook: persistent blocks per launch of the direct 3x3 kernel (csrc/dconv3_b3.hip; 0 restores the default 512 = two per CU) */
int catseg_debug_set_dconv3_blocks(int blocks);
/* A/B hook: 1 = the wave-specialised variant of the direct 3x3 kernel (4 compute + 4 helper waves per block), 0 = uniform waves,
 * -1 (default) = the library's choice per channel count (specialised for 192 / 384 channels) */
int catseg_debug_set_dconv3_spec(int on);
/* A/B hook: 1 = 96-channel layers on 4 x 32-pixel tiles with uniform waves instead of the default 4 x 16 tiles with specialised waves */
int catseg_debug_set_dconv3_alt96(int on);
/* tuning hook: blocks per launch of the direct backward-weight kernel (csrc/dwgrad3_b3.hip; 0 restores the default 512) */
int catseg_debug_set_dwgrad3_blocks(int blocks);
/* persistent blocks of the planes kernel csrc/dconv3_pl.hip (default 512 = two per CU) */
/* tuning hook: blocks of the 96+ channel backward-weight kernel on planes (0 = default 512; 768 = three per CU: faster standalone, more slab traffic, neutral in the step) */
int catseg_debug_set_dwgrad3_pl_blocks(int blocks);
int catseg_debug_set_dconv3_pl_slots(int slots);
/* tuning hook: bit mask of channel counts (2: 96, 4: 192, 8: 384) whose planes kernel runs in the two-tiles-per-block form (eight
 * compute waves sharing one stream of weights, one block per CU) */
int catseg_debug_set_dconv3_pl_pair(int mask);
/* occupancy queries (blocks per CU the runtime grants the planes kernels; needs a GPU): the designs assume two, or one for pair != 0 */
int catseg_debug_dconv3_pl_occupancy(int C, int pair);
int catseg_debug_dwgrad3_pl_occupancy(int C);

/* test hook: 1 = igemm_h2w8_kernel always runs its general (predicated) epilogue; 0 (default) = the full-row-tile epilogue where the launch allows */
int catseg_debug_set_h2w_slow_epilogue(int on);

/* tuning hook: bf16x3 block tile: 0 = heuristic, 1 = 256x256, 2 = 256x128, 3 = 128x256, 4 = 256x192, 5 = 256x96, 6 = 256x64 */
int catseg_debug_set_b3_tile(int t);
/* planner query: which tile / split count would the library pick for this convolution?
 * op: 0 = forward, 1 = backward-data (for stride > 1: the launch of input-pixel parity class (0, 0)),
 *     2 = backward-weight.  out[0..4] = mi, ni, form (0 = 32x32x2 tiles (64 mi) x (64 ni); 1 / 2 = 16x16x4 tiles narrow
 *     in N / M; 3 / 4 = 16 / 32 wide), splits, direct (1 = the direct backward-weight kernel handles it). */
int catseg_debug_plan_conv(const catseg_conv_desc* d, int op, int* out);

/* A/B hook: 1 (default) = catseg_bilinear_bwd as one launch with the intermediate row in LDS where the layout allows 16-byte loads
 * (csrc/pointwise.hip: bilinear_bwd_fused_kernel), 0 = the two separable passes through the workspace.  Bit-identical results. */
int catseg_debug_set_bilinear_bwd_fused(int on);
/* A/B hook: waves per block of the f16x2 forward / backward-data kernel of the head layers (csrc/igemm_f16x2.hip): 8 (default) =
 * igemm_h2w8_kernel, two waves per SIMD; 4 = igemm_h2w_kernel, one 512-register wave per SIMD.  Bit-identical results. */
int catseg_debug_set_h2w_waves(int waves);

/* csrc/pconv1.hip: blocks of a backward-weight launch of the pointwise kernels (default 256: one per CU) */
int catseg_debug_set_pconv1_wgrad_blocks(int n);

#ifdef __cplusplus
}
#endif
#endif /* CATSEG_DEBUG_H */
