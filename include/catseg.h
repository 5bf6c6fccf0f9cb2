/*
 * catseg.h — C ABI of libcatseg_hip.so: the MI355X (gfx950) hot path of the
 * cataract-segmentation trainer.
 *
 * The reference (RViMLab/MICCAI2021_Cataract_semantic_segmentation) is 100 % Python and
 * has no FFI: its hot path is the list of ATen ops dispatched from models/*.py and
 * losses/*.py.  Each entry point below replaces one such ATen call site (cited as
 * file:line relative to the reference tree).  The boundary is plain C: raw device
 * pointers, explicit sizes / strides, a hipStream_t passed as void*, int status.
 *
 * Conventions
 *  - activations are NHWC fp32: pixel p = (b*H + y)*W + x, element (p, c) at base[p*ld + c],
 *    ld >= C and ld % 4 == 0, base 16-byte aligned ("ld" lets a tensor live inside a
 *    wider concat buffer without a copy);
 *  - conv weights are OHWI fp32 (a torch [O,I,kh,kw] tensor in channels_last memory format);
 *  - every function is asynchronous on `stream`, never allocates, never synchronises;
 *  - return value: 0 = ok, otherwise a CATSEG_E* code; catseg_last_error() gives a message.
 */
#ifndef CATSEG_H
#define CATSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CATSEG_OK 0
#define CATSEG_EINVAL 1   /* bad argument (shape / alignment)  */
#define CATSEG_EHIP 2     /* a HIP runtime call failed          */
#define CATSEG_EWORKSPACE 3

typedef void* catseg_stream_t; /* hipStream_t */

const char* catseg_last_error(void);
int catseg_version(void);

/* ---- convolution as implicit GEMM on v_mfma_f32_32x32x2_f32 ------------------------------
 * replaces F.conv2d behind nn.Conv2d at models/OCR.py:72-97,200-235,308-313 and
 * models/DeepLabv3Plus.py:92-105,145-156, models/HRNetv2.py:23-33 etc. and every
 * torchvision ResNet conv (models/OCR.py:58-61). */
typedef struct {
  int B, H, W, Cin;        /* input  NHWC dims                                   */
  int Ho, Wo, Cout;        /* output NHWC dims                                   */
  int kh, kw, stride, pad, dil;
  int ldx, ldy;            /* floats per pixel of x / y buffers                  */
  int stem4;               /* 1: 7x7/2 stem on a 4-channel-padded image, weights packed [O][kh][8][4] */
  int groups;              /* 0 or 1 = dense; g > 1 = grouped (forward only: ResNeXt inference, models/ResNeXt.py:46-60);
                              weights [Cout][kh][kw][Cin/g], Cin/g a multiple of 4 */
} catseg_conv_desc;

/* y[p, o] = sum_{ky,kx,c} x[pix(p,ky,kx), c] * w[o,ky,kx,c] (+ bias[o]); columns
 * [Cout, zero_to) of y are written as zeros (zero_to <= ldy, 0 = none). */
int catseg_conv2d_fwd(const catseg_conv_desc* d, const float* x, const float* w, const float* bias,
                      float* y, int zero_to, catseg_stream_t stream);
/* training forward of conv -> BatchNorm (models/OCR.py:72-76 etc.): catseg_conv2d_fwd whose epilogue also writes per-(M-tile,
 * channel) BatchNorm partials [n_tiles][3][Cout] (K, s1, s2) for catseg_bn_finalize -- no separate statistics pass over y.
 * *tile_rows == 0 on return: this layer's tile form has no fused statistics; use catseg_bn_train_stats. */
int catseg_conv2d_fwd_bnstats(const catseg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                              int zero_to, float* bn_part, size_t bn_part_floats, int* tile_rows, int* n_tiles,
                              catseg_stream_t stream);
/* inference: y = act(conv(x, w) + bias (+ residual)), act = relu if relu != 0, in one kernel.  With
 * catseg_fold_bn this is Conv2d + eval-mode BatchNorm2d + residual add + ReLU of a ResNet / UPerNet block. */
int catseg_conv2d_fwd_fused(const catseg_conv_desc* d, const float* x, const float* w, const float* bias,
                            const float* residual, int ldr, int relu, float* y, catseg_stream_t stream);
/* w'[o,:] = w[o,:] * gamma[o]/sqrt(rv[o]+eps), b'[o] = beta[o] + (b[o] - rm[o]) * gamma[o]/sqrt(rv[o]+eps);
 * per_out = floats per output channel of w (kh*kw*Cin, or 7*32 for the packed stem) */
int catseg_fold_bn(const float* w, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, float eps, int O, int per_out, float* w_folded, float* bias_folded,
                   catseg_stream_t stream);
/* dx[q, c] (+)= sum_{ky,kx,o} dy[opix(q,ky,kx), o] * w[o,ky,kx,c]  (autograd of the above) */
int catseg_conv2d_bwd_data(const catseg_conv_desc* d, const float* dy, const float* w, float* dx,
                           int accumulate, catseg_stream_t stream);
/* dw[o,ky,kx,c] = sum_p dy[p, o] * x[pix(p,ky,kx), c]; dbias[o] = sum_p dy[p,o] (may be NULL).
 * workspace: catseg_conv2d_bwd_weight_workspace() bytes (split-reduction slabs). */
size_t catseg_conv2d_bwd_weight_workspace(const catseg_conv_desc* d);
int catseg_conv2d_bwd_weight(const catseg_conv_desc* d, const float* x, const float* dy, float* dw,
                             float* dbias, void* workspace, size_t workspace_bytes,
                             catseg_stream_t stream);

/* ---- batched GEMM for the OCR gather / object attention (torch.matmul at
 * models/OCR.py:167,266,274).  C[z] (M x N, ldc) = op(A[z]) * op(B[z]):
 *   layout "NT": A is M x K (lda), B is N x K (ldb)      (K contiguous in both)
 *   layout "NN": A is M x K (lda), B is K x N (ldb)
 *   layout "TN": A is K x M (lda), B is K x N (ldb)      (reduction over rows) */
#define CATSEG_GEMM_NT 0
#define CATSEG_GEMM_NN 1
#define CATSEG_GEMM_TN 2
/* second stage of a K-split reduction: out[b][i] (+)= sum_s slabs[b][s][i] in a fixed order (slabs [batch][splits][n], n % 4 == 0).  The
 * OCR head's reductions over all N = H * W pixels into a K x C result (models/OCR.py:158-170 spatial gather, the value / key gradients of
 * :266-274) run as catseg_gemm_batched over batch * splits row chunks followed by this sum */
int catseg_sum_slabs(const float* slabs, float* out, long long n, int splits, int batch, int accumulate, catseg_stream_t stream);
int catseg_gemm_batched(int layout, int batch, int M, int N, int K, const float* A, int lda,
                        long long strideA, const float* Bm, int ldb, long long strideB, float* C,
                        int ldc, long long strideC, int zero_to, int accumulate,
                        catseg_stream_t stream);

/* ---- split-precision convolution on the bf16 matrix cores (csrc/igemm_bf16x3.hip): every fp32 operand is split exactly into
 * three bf16 planes (catseg_split3), six bf16 MFMA partial products per block reproduce the fp32 product to ~2^-23;
 * 2.7x the fp32-matrix peak.  Same F.conv2d call sites as catseg_conv2d_fwd / _bwd_data; results agree with the fp32 kernels
 * to ~1e-6 relative (not bit-identical). */
size_t catseg_split3_elems(long long rows, int C);   /* 16-bit elements per plane: rows * roundup(C, 8) */
int catseg_split3(const float* x, int ld, long long rows, int C, void* planes, catseg_stream_t stream);
int catseg_split3_weight_t(const float* w, int O, int taps, int Cin, void* planes, catseg_stream_t stream);
int catseg_conv2d_fwd_bf16x3(const catseg_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                             float* y, int zero_to, catseg_stream_t stream);
int catseg_conv2d_fwd_bf16x3_bnstats(const catseg_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                                     float* y, int zero_to, float* bn_part, size_t bn_part_floats, int* tile_rows, int* n_tiles,
                                     catseg_stream_t stream);
int catseg_conv2d_bwd_data_bf16x3(const catseg_conv_desc* d, const void* dy_planes, const void* wt_planes, float* dx,
                                  int accumulate, catseg_stream_t stream);
/* dw = backward-weight from pre-split planes (x: C = Cin, dy: C = Cout); workspace = split-reduction slabs */
size_t catseg_conv2d_bwd_weight_bf16x3_workspace(const catseg_conv_desc* d);
int catseg_conv2d_bwd_weight_bf16x3(const catseg_conv_desc* d, const void* x_planes, const void* dy_planes, float* dw,
                                    void* workspace, size_t workspace_bytes, catseg_stream_t stream);
/* BLOCKED operand planes for the large layers (>= 193 output columns): activations [3][ceil(C/16)][rows][16], weights
 * [3][K/16][N][16] -- the 256 rows of one K-step form one contiguous run, so that every LDS-DMA instruction of the 256 x 256
 * kernel reads whole cache lines (the [rows][C] planes above gave each lane pair its own line and bounded that kernel by the
 * vector L1's line rate).  catseg_split3_blocked writes, from ONE pass over x, the blocked planes and -- if planar_planes is not
 * null -- the catseg_split3 layout as well (the backward-weight kernel keeps reading that one). */
size_t catseg_split3_blocked_elems(long long rows, int C);   /* 16-bit elements of ALL three planes: 3 * roundup(C, 16) * rows */
int catseg_split3_blocked(const float* x, long long rows, int C, int ld, void* blocked_planes, void* planar_planes,
                          catseg_stream_t stream);
int catseg_split3_weight_blocked(const float* w, int O, int taps, int Cin, void* planes, catseg_stream_t stream);    /* Cin % 16 == 0 */
int catseg_split3_weight_t_blocked(const float* w, int O, int taps, int Cin, void* planes, catseg_stream_t stream);  /* [taps*roundup(O,16)/16][Cin][16] */
/* forward (Cin % 16 == 0; bn_part may be NULL: no BatchNorm partials) and stride-1 backward-data from blocked planes */
int catseg_conv2d_fwd_bf16x3_blocked(const catseg_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                                     float* y, int zero_to, float* bn_part, size_t bn_part_floats, int* tile_rows, int* n_tiles,
                                     catseg_stream_t stream);
/* inference: y = act(conv + bias (+ residual)) in one kernel, the bf16x3 counterpart of catseg_conv2d_fwd_fused */
int catseg_conv2d_fwd_fused_bf16x3_blocked(const catseg_conv_desc* d, const void* x_planes, const void* w_planes, const float* bias,
                                           const float* residual, int ldr, int relu, float* y, catseg_stream_t stream);
int catseg_conv2d_bwd_data_bf16x3_blocked(const catseg_conv_desc* d, const void* dy_planes, const void* wt_planes, float* dx,
                                          int accumulate, catseg_stream_t stream);
/* dbias[o] = sum_p dy[p, o] (the bias-gradient part of catseg_conv2d_bwd_weight on its own); workspace >= 256 * C floats */
int catseg_bias_grad(const float* dy, int ld, long long rows, int C, float* dbias, void* workspace, size_t workspace_bytes,
                     catseg_stream_t stream);

/* ---- two-plane fp16 split precision (csrc/igemm_f16x2.hip): the same F.conv2d call sites as the bf16x3 blocked entry points above
 * (the OCR / auxiliary head convolutions 3x3 720 -> 512 and the 1x1 1024 -> 512 of models/OCR.py:88-104, UPerNet's fusion layers), half
 * the matrix work.  Every operand tensor is scaled by 2^e, e = 14 - floor(log2(max|x|)), and split into h = fp16(x 2^e),
 * l = fp16(x 2^e - h) (22 significant bits); the kernel accumulates hh + hl + lh in fp32 and scales the result by 2^-(e_x + e_w).
 *   `scale`: 8 bytes of DEVICE memory per operand, {uint32 bits of max|x|, int32 e}, written by the split call (two launches: amax,
 *   split) and read by the convolution -- no host round trip. */
size_t catseg_split2h_blocked_elems(long long rows, int C);   /* [2][ceil(C/16)][rows][16]: forward / backward-data operand */
size_t catseg_split2h_planar_elems(long long rows, int C);    /* [2][rows][roundup(C, 8)]: backward-weight operand */
int catseg_split2h(const float* x, long long rows, int C, int ld, void* blocked_planes, void* planar_planes, void* scale,
                   catseg_stream_t stream);                   /* either layout may be NULL; one pass over x for both */
/* catseg_split2h without the pass over x that finds max|x|: the maximum over up to four amax records the producers of x (or of its channel
 * slices) left (null = unused); a record may be an upper bound */
int catseg_split2h_bound(const float* x, long long rows, int C, int ld, void* blocked_planes, void* planar_planes, void* scale,
                         const void* rec0, const void* rec1, const void* rec2, const void* rec3, catseg_stream_t stream);
int catseg_split2h_weight_blocked(const float* w, int O, int taps, int Cin, void* planes, void* scale, catseg_stream_t stream);
/* blocked planes of the channel concatenation of up to four bilinearly resized tensors (F.interpolate(..., mode='bilinear', align_corners=False)
 * + torch.cat: the HRNet head input, models/HRNetv2.py:505-508) WITHOUT the fp32 concatenation: xs[i] NHWC [B][Hs[i]][Ws[i]][Cs[i]] (row stride
 * lds[i], Cs[i] % 16 == 0; a source of the output's size is copied), records[i] its amax record; planes [2][sum Cs / 16][B Ho Wo][16]. */
int catseg_concat_bilinear_split2h(int nsrc, const float* const* xs, const int* lds, const int* Hs, const int* Ws, const int* Cs,
                                   const void* const* records, int B, int Ho, int Wo, void* blocked_planes, void* scale, catseg_stream_t stream);
int catseg_split2h_weight_t_blocked(const float* w, int O, int taps, int Cin, void* planes, void* scale, catseg_stream_t stream);
int catseg_conv2d_fwd_f16x2_blocked(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* w_planes,
                                    const void* w_scale, const float* bias, float* y, int zero_to, float* bn_part, size_t bn_part_floats,
                                    int* tile_rows, int* n_tiles, catseg_stream_t stream);
int catseg_conv2d_fwd_fused_f16x2_blocked(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* w_planes,
                                          const void* w_scale, const float* bias, const float* residual, int ldr, int relu, float* y,
                                          catseg_stream_t stream);
int catseg_conv2d_bwd_data_f16x2_blocked(const catseg_conv_desc* d, const void* dy_planes, const void* dy_scale, const void* wt_planes,
                                         const void* wt_scale, float* dx, int accumulate, catseg_stream_t stream);
size_t catseg_conv2d_bwd_weight_f16x2_workspace(const catseg_conv_desc* d);
int catseg_conv2d_bwd_weight_f16x2(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* dy_planes,
                                   const void* dy_scale, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream);
/* the same from the BLOCKED planes of catseg_split2h (the operand layout of the forward / backward-data kernels): one plane set per tensor
 * serves all three directions of a layer (the backward of F.conv2d at models/OCR.py:72-89, 326-333). */
int catseg_conv2d_bwd_weight_f16x2_blocked(const catseg_conv_desc* d, const void* x_planes, const void* x_scale, const void* dy_planes,
                                           const void* dy_scale, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream);

/* ---- the first stem convolution of HRNet: nn.Conv2d(3, 64, 3, stride 2, padding 1) on the image (models/HRNetv2.py:281-283), exact fp32, HBM-bound
 * direct kernels (csrc/stem3.hip).  x is addressed through element strides (sb, sc, sy, sx) for (batch, channel, row, column): NCHW or NHWC-4,
 * no repack pass.  w / dw: OHWI [64][3][3][3].  y / dy: NHWC rows of ldy floats.
 *   catseg_stem3_fwd: y = F.conv2d(x, w, bias, 2, 1); bn_part != NULL: catseg_stem3_partial_rows(B, H, W) rows [row][3][64] of BatchNorm
 *                     partials (K, sum(v - K), sum((v - K)^2)) with pixel counts in bn_counts, for catseg_bn_finalize_counts.
 *   catseg_stem3_bwd_weight: dw = the weight gradient of the same call (autograd of F.conv2d); workspace catseg_stem3_wgrad_workspace(). */
int catseg_stem3_supported(int H, int W, int Cout);
int catseg_stem3_partial_rows(int B, int H, int W);
size_t catseg_stem3_wgrad_workspace(void);
int catseg_stem3_fwd(const float* x, long long sb, long long sc, long long sy, long long sx, int B, int H, int W, const float* w, const float* bias,
                     float* y, int ldy, float* bn_part, int* bn_counts, catseg_stream_t stream);
int catseg_stem3_bwd_weight(const float* x, long long sb, long long sc, long long sy, long long sx, int B, int H, int W, const float* dy, int lddy,
                            float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream);

/* ---- the first convolution of the torchvision ResNet stem: nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False) on the image (the resnet50 / resnet101
 * backbones built at models/OCR.py:58-61 and models/DeepLabv3Plus.py:32-38 of the reference; torchvision's `conv1`), training forward as a direct
 * kernel whose 147 products per output are accumulated in fp64 and rounded once (csrc/stem7.hip).  Arguments as catseg_stem3_fwd; w: OHWI [64][7][7][3].
 *   catseg_stem7_fwd: y = F.conv2d(x, w, bias, 2, 3); bn_part != NULL: catseg_stem7_partial_rows(B, H, W) rows of BatchNorm partials + pixel counts. */
int catseg_stem7_supported(int H, int W, int Cout);
int catseg_stem7_partial_rows(int B, int H, int W);
int catseg_stem7_fwd(const float* x, long long sb, long long sc, long long sy, long long sx, int B, int H, int W, const float* w, const float* bias,
                     float* y, int ldy, float* bn_part, int* bn_counts, catseg_stream_t stream);

/* ---- direct 3x3 / stride 1 / pad 1 convolution in split precision (csrc/dconv3_b3.hip) for the HRNet trunk: the BasicBlock
 * convolutions conv3x3(planes, planes) at models/HRNetv2.py:22-25,41-44 (Cin = Cout = C in {48, 96}; catseg_dconv3_supported).
 * Same F.conv2d call sites and same arithmetic as catseg_conv2d_fwd_bf16x3 (three exact bf16 planes per fp32 operand, six bf16
 * MFMA products, fp32 accumulation), but the activation is read as fp32 and split inside the kernel: a block loads the halo
 * tile of its pixel tile once and forms all nine taps from it.
 *   catseg_dconv3_prep: OHWI fp32 weights [C][3][3][C] -> the kernel's pre-split weight image (catseg_dconv3_wimg_bytes bytes);
 *                       backward_data != 0: the transposed, tap-mirrored bank, with which the same kernel run on dy computes dx.
 *   catseg_dconv3:      y[p, o] (+)= sum_{ky,kx,c} x[pix(p,ky,kx), c] * w[o,ky,kx,c] (+ bias[o]).  bn_part != NULL: BatchNorm partials
 *                       [n_tiles][3][C] (tile mean, s1, s2) + bn_counts[n_tiles] (valid pixels of each tile; n_tiles =
 *                       catseg_dconv3_tiles) for catseg_bn_finalize_counts. */
int catseg_dconv3_supported(int C);
size_t catseg_dconv3_wimg_bytes(int C);
int catseg_dconv3_tiles(int C, int B, int H, int W, int* tile_h, int* tile_w);
int catseg_dconv3_prep(const float* w, int C, int backward_data, void* wimg, catseg_stream_t stream);
/* the weight images of ALL direct-kernel layers of a network in one launch (the parameters live in one flat buffer): `entries` is a
 * DEVICE array of n records {int64 weight offset in floats from `flat`; int64 image offset in bytes from wimg_base; int32 C; int32 KC;
 * int32 NT; int32 backward_data} with KC / NT from catseg_dconv3_layout(C) */
int catseg_dconv3_layout(int C, int* kc, int* nt);
int catseg_dconv3_prep_batch(const float* flat, int n, const void* entries, void* wimg_base, catseg_stream_t stream);
int catseg_dconv3(int B, int H, int W, int C, const float* x, int ldx, const void* wimg, const float* bias, float* y, int ldy,
                  int accumulate, float* bn_part, size_t bn_part_floats, int* bn_counts, catseg_stream_t stream);
/* backward-data of a BasicBlock's SECOND convolution fused with the first pass of the backward of the relu(bn1(q)) that produced
 * its input (models/HRNetv2.py:36-47: out = relu(bn1(conv1(x))); conv2(out)): g = (dy (*) w^T) where relu(bn1(q)) > 0 (the mask is
 * recomputed from q, stats = [mean(C), invstd(C)], gamma, beta with the forward's exact expression), and part[n_tiles][2][C] = per
 * tile sum of g and of g * xhat -- what the first pass of catseg_bn_backward computes from (dz, q).  catseg_bn_backward_pre finishes. */
int catseg_dconv3_bnbwd(int B, int H, int W, int C, const float* dy, int lddy, const void* wimg_bwd, float* g, int ldg, const float* q,
                        int ldq, const float* stats, const float* gamma, const float* beta, float* part, size_t part_floats,
                        catseg_stream_t stream);
/* The same direct kernels on TWO fp16 planes and THREE products (csrc/dconv3_f16x2.hip = dconv3_b3.hip compiled with DC_H2; the
 * arithmetic of catseg_conv2d_fwd_f16x2_blocked).  The activation is split inside the kernel, so its power-of-two prescale comes from a
 * DEVICE amax record of CATSEG_AMAX_RECORD_BYTES = 2048 bytes (16 uint32 slots 128 bytes apart; max over the slots = bits of max|x|
 * over the whole tensor) that the PRODUCER of x accumulated (catseg_bn_apply_amax, catseg_add_n_act_amax, catseg_bn_backward_amax /
 * _pre_amax: one fire-and-forget atomicMax per block into slot blockIdx % 16; the caller zeroes the record before the producer
 * runs).  A record value above the true maximum is safe, one below it overflows fp16.
 *   catseg_dconv3_f16x2_prep_batch: as catseg_dconv3_prep_batch (same entry records) + `records`, n x 8 bytes: per weight image
 *                                   {bits of max|w|, exponent}, computed on the device (memset + amax launch + image launch). */
#define CATSEG_AMAX_RECORD_BYTES 2048
size_t catseg_dconv3_f16x2_wimg_bytes(int C);
int catseg_dconv3_f16x2_prep_batch(const float* flat, int n, const void* entries, void* wimg_base, void* records, catseg_stream_t stream);
int catseg_dconv3_f16x2(int B, int H, int W, int C, const float* x, int ldx, const void* x_record, const void* wimg, const void* w_record,
                        const float* bias, float* y, int ldy, int accumulate, float* bn_part, size_t bn_part_floats, int* bn_counts,
                        catseg_stream_t stream);
int catseg_dconv3_bnbwd_f16x2(int B, int H, int W, int C, const float* dy, int lddy, const void* dy_record, const void* wimg_bwd,
                              const void* w_record, float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma,
                              const float* beta, float* part, size_t part_floats, catseg_stream_t stream);
/* backward-weight on two fp16 planes (csrc/dwgrad3_f16x2.hip = dwgrad3_b3.hip compiled with DW_H2): both operands' amax records;
 * workspace = catseg_dwgrad3_workspace */
int catseg_dwgrad3_f16x2(int B, int H, int W, int C, const float* x, int ldx, const void* x_record, const float* dy, int ldy,
                         const void* dy_record, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream);
/* backward-weight of the same layers (csrc/dwgrad3_b3.hip): dw[o,ky,kx,c] = sum_p dy[p, o] * x[pix(p,ky,kx), c], C in {48, 96, 192, 384};
 * x and dy are read as fp32 and split inside the kernel, fragments by transposed LDS reads, per-block partial sums in
 * `workspace` (catseg_dwgrad3_workspace bytes) added in a fixed order */
int catseg_dwgrad3_supported(int C);
size_t catseg_dwgrad3_workspace(int B, int H, int W, int C);
int catseg_dwgrad3(int B, int H, int W, int C, const float* x, int ldx, const float* dy, int ldy, float* dw, void* workspace,
                   size_t workspace_bytes, catseg_stream_t stream);

/* ---- FCN-8s pieces (models/FCN.py:7-61 of the reference): F.max_pool2d(x, 2) with its argmax (idx: uint8 [B, H/2, W/2, C]) and backward
 * (models/FCN.py:44-53); catseg_bias_rows: out[r][0..C) = bias -- the rows nn.ConvTranspose2d's bias initialises before the transposed
 * convolution (= catseg_conv2d_bwd_data with the same weight tensor, models/FCN.py:35-38) accumulates into them */
int catseg_maxpool2x2_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx, int B, int H, int W, int C, catseg_stream_t stream);
int catseg_maxpool2x2_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx, int B, int H, int W, int C, catseg_stream_t stream);
int catseg_bias_rows(const float* bias, float* out, int ld, long long rows, int C, catseg_stream_t stream);

/* ---- fp16 x 2 OPERAND PLANES written by the producer of a tensor, and the direct 3x3 kernels that stream them (round 4;
 * csrc/planes.h, csrc/dconv3_pl.hip).  Replaces, for the same reference layers as catseg_dconv3_f16x2 (conv3x3(planes, planes) of
 * models/HRNetv2.py:22-65), the in-kernel fp32 -> 2 x fp16 split: planes = [plane h / l][C / 8 channel groups][pixel][8] fp16,
 * catseg_planes_bytes(rows, C) bytes, written with the exponent e that the tensor's amax record holds in word 1 (xs = x * 2^e).
 *   catseg_planes_from_f32:  standalone producer (tests; tensors whose producing kernel writes no planes).  compute_amax != 0: the record
 *                            is zeroed and max|x| taken here; else its 16 amax slots are taken as they are (a larger value is safe).
 *   catseg_dconv3_pl:        catseg_dconv3_f16x2 on planes.  BatchNorm partials PER WAVE: catseg_dconv3_pl_rows(C, B, H, W) rows of
 *                            [3][C] floats + one int32 count per row, for catseg_bn_finalize_counts.  out_record (may be NULL): max|y| is
 *                            folded into its amax slots (the BatchNorm that follows derives the exponent of ITS output planes from it).
 *   catseg_dconv3_pl_bnbwd:  catseg_dconv3_bnbwd_f16x2 on planes of dy; part = [rows][2][C]; max|g| to out_record. */
int catseg_dconv3_pl_supported(int C);
int catseg_dconv3_pl_rows(int C, int B, int H, int W);
size_t catseg_planes_bytes(long long rows, int C);
int catseg_planes_from_f32(const float* x, int ld, long long rows, int C, void* record, void* planes, int compute_amax, catseg_stream_t stream);
int catseg_dconv3_pl(int B, int H, int W, int C, const void* planes, const void* planes_record, const void* wimg, const void* w_record,
                     const float* bias, float* y, int ldy, int accumulate, float* bn_part, size_t bn_part_floats, int* bn_counts,
                     void* out_record, catseg_stream_t stream);
int catseg_dconv3_pl_bnbwd(int B, int H, int W, int C, const void* dy_planes, const void* dy_record, const void* wimg_bwd, const void* w_record,
                           float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma, const float* beta, float* part,
                           size_t part_floats, void* out_record, catseg_stream_t stream);

/* PRODUCERS of planes: the kernels that write a trunk activation / gradient write its planes in the same pass (csrc/norm.hip,
 * csrc/pointwise.hip).  The exponent is fixed BEFORE the pass from a bound of max| |:
 *   catseg_bn_finalize_counts_bound: catseg_bn_finalize_counts, and z_record[2] = max_c |scale_c| (max|y| + |mean_c|) + |beta_c| (max|y| from
 *                                    y_record, the record catseg_dconv3_pl's epilogue filled)                  [nn.BatchNorm2d training forward]
 *   catseg_bn_apply_planes:          catseg_bn_apply writing planes (+ z itself when z != NULL); exponent from z_record's bound + max|residual|
 *   catseg_add_n_act_planes:         catseg_add_n_act writing out and its planes; exponent from the sum of the terms' max| |  [HRNet fuse sum]
 *   catseg_bn_backward_planes / _pre_planes: catseg_bn_backward / catseg_bn_backward_pre with dy as planes ONLY; bound
 *                                    |gamma invstd| (max|g| + |mean g| + max|xhat| |mean g xhat|) per channel       [autograd of BatchNorm2d] */
int catseg_bn_finalize_counts_bound(const float* partials, int n_blocks, const int* counts, long long rows, int C, const float* gamma,
                                    const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* stats_out,
                                    float* scale, const void* y_record, void* z_record, catseg_stream_t stream);
int catseg_bn_apply_planes(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual, int ldr,
                           const void* residual_record, float* z, int ldz, void* z_planes, long long rows, int C, int relu, void* z_record,
                           catseg_stream_t stream);
int catseg_add_n_act_planes(const float* const* in, const int* ld, const void* const* term_records, int n, float* out, int ldo, void* out_planes,
                            long long rows, int C, int relu, void* out_record, catseg_stream_t stream);
int catseg_bn_backward_planes(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* stats, const float* gamma,
                              const float* beta, long long rows, int C, int relu, void* dy_planes, void* dy_record, void* g_record,
                              const void* y_record, float* dgamma, float* dbeta, float* dres, int lddres, int dres_accumulate, void* workspace,
                              size_t workspace_bytes, catseg_stream_t stream);
int catseg_bn_backward_pre_planes(const float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma, const float* partials,
                                  int n_blocks, long long rows, int C, void* dq_planes, void* dq_record, const void* g_record,
                                  const void* y_record, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                  catseg_stream_t stream);
/* catseg_bn_backward with dy written ONLY as the blocked fp16 x 2 planes of catseg_split2h (dy_planes: catseg_split2h_blocked_elems(rows, C)
 * halves; dy_scale: its 8-byte record {bits of the bound, e}) -- for a layer whose backward-weight / backward-data run
 * catseg_conv2d_bwd_weight_f16x2_blocked / catseg_conv2d_bwd_data_f16x2_blocked (the BatchNorm behind the head convolutions, models/OCR.py:72-89,
 * 326-333).  dbias (may be null) = column sums of dy (gradient of a convolution bias in front of the BatchNorm).  No residual branch; C % 64 == 0.
 * g_record / y_record / dy_record: three zeroed amax records; they receive max|masked gradient|, max|y| and the bound dy's exponent came from. */
size_t catseg_bn_backward_h2_workspace(long long rows, int C);
int catseg_bn_backward_h2(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* stats, const float* gamma,
                          const float* beta, long long rows, int C, int relu, void* dy_planes, void* dy_scale, float* dgamma, float* dbeta,
                          float* dbias, void* g_record, void* y_record, void* dy_record, void* workspace, size_t workspace_bytes,
                          catseg_stream_t stream);

/* The K-class classifier of a segmentation head fused with the BatchNorm + ReLU in front of it (csrc/headfuse.h; replaces nn.BatchNorm2d + ReLU +
 * the 1 x 1 nn.Conv2d of models/OCR.py:72-74 (interm_prediction_head[1..4]) and :97 with conv_bn_dropout[1..2], and their autograd): the
 * normalised activation z and its gradient exist in registers only.
 *   catseg_head_fwd       logits[rows][ldl] = relu((y - mean) * scale + beta) Wh^T + bh, columns [K, zero_to) zeroed; Wh [K][C], K <= 32,
 *                         C % 32 == 0, C <= 512; mean / scale as catseg_bn_finalize left them
 *   catseg_head_backward  from dl [rows][lddl] (lddl >= 32): dy as the blocked planes + scale record of catseg_bn_backward_h2 (same three zeroed
 *                         amax records), dgamma / dbeta of the BatchNorm, dbias (may be null) = column sums of dy, dwh [K][C] and dbh [K]
 *                         (may be null) of the classifier -- all WRITTEN; C % 64 == 0; workspace = catseg_head_backward_workspace bytes;
 *                         every reduction in a fixed order (deterministic) */
int catseg_head_fwd(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* wh, const float* bh, int K,
                    long long rows, int C, float* logits, int ldl, int zero_to, catseg_stream_t stream);
size_t catseg_head_backward_workspace(long long rows, int C);
int catseg_head_backward(const float* dl, int lddl, const float* y, int ldy, const float* stats, const float* gamma, const float* beta,
                         const float* wh, int K, long long rows, int C, void* dy_planes, void* dy_scale, float* dgamma, float* dbeta, float* dbias,
                         float* dwh, float* dbh, void* g_record, void* y_record, void* dy_record, void* workspace, size_t workspace_bytes,
                         catseg_stream_t stream);

/* catseg_dwgrad3_f16x2 on producer-written planes of BOTH operands (csrc/dwgrad3_pl.hip): dw[o][ky][kx][c] = sum_px dy[px][o] x[px + tap][c]
 * for the trunk widths 48 / 96 / 192 / 384 (autograd of F.conv2d, models/HRNetv2.py:22-65); workspace = catseg_dwgrad3_pl_workspace bytes
 * (slabs of partial sums, added in a fixed order: deterministic) */
int catseg_dwgrad3_pl_supported(int C);
size_t catseg_dwgrad3_pl_workspace(int B, int H, int W, int C);
int catseg_dwgrad3_pl(int B, int H, int W, int C, const void* x_planes, const void* x_record, const void* dy_planes, const void* dy_record,
                      float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream);

/* (tuning / measurement hooks live in catseg_debug.h: they are process-global and not part of the product surface) */

/* ---- BatchNorm (+ReLU, +residual) — nn.BatchNorm2d/ReLU at e.g. models/OCR.py:74-75,
 * torchvision Bottleneck, models/DeepLabv3Plus.py:98-104.  rows = B*H*W pixels. */
/* batch statistics + running-stat update.  stats_out = [mean(C), invstd(C)] (biased variance),
 *   scale = gamma*invstd.  running_mean/var may be NULL (no update; unbiased variance, momentum
 *   as nn.BatchNorm2d).  workspace >= catseg_bn_workspace(rows, C). */
size_t catseg_bn_workspace(long long rows, int C);
int catseg_bn_train_stats(const float* y, long long rows, int C, int ldy, const float* gamma, float eps,
                          float momentum, float* running_mean, float* running_var, float* stats_out,
                          float* scale, void* workspace, size_t workspace_bytes, catseg_stream_t stream);
/* merge of [n_blocks][3][C] partials (per row block: K, sum(y - K), sum((y - K)^2)) into the batch statistics: the second half
 * of catseg_bn_train_stats, fed by the convolution epilogues above */
int catseg_bn_finalize(const float* partials, int n_blocks, long long rows_per_block, long long rows, int C, const float* gamma,
                       float eps, float momentum, float* running_mean, float* running_var, float* stats_out, float* scale,
                       catseg_stream_t stream);
/* the same merge for blocks with individual row counts (2-D pixel tiles with ragged edges: catseg_dconv3) */
int catseg_bn_finalize_counts(const float* partials, int n_blocks, const int* counts, long long rows, int C, const float* gamma,
                              float eps, float momentum, float* running_mean, float* running_var, float* stats_out, float* scale,
                              catseg_stream_t stream);
/* eval mode: scale = gamma / sqrt(running_var + eps) (use with mean = running_mean) */
int catseg_bn_eval_scale(int C, const float* gamma, const float* running_var, float eps, float* scale,
                         catseg_stream_t stream);
/* z = act((y - mean) * scale + beta (+ residual)),  act = relu if relu != 0 */
int catseg_bn_apply(const float* y, int ldy, const float* mean, const float* scale, const float* beta,
                    const float* residual, int ldr, float* z, int ldz, long long rows, int C, int relu,
                    catseg_stream_t stream);
/* backward of the fused op.  g = dz * (z > 0 if relu).  Produces dgamma, dbeta, dy and, when
 * dres != NULL, the residual-branch gradient (dres (+)= g if dres_accumulate).
 * z may be NULL when the forward had no residual branch: the ReLU mask is then recomputed from y, gamma, beta and
 * the saved statistics with the forward's exact expression (one tensor read less in both passes).
 * workspace >= catseg_bn_workspace(rows, C). */
int catseg_bn_backward(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                       const float* stats, const float* gamma, const float* beta, long long rows, int C, int relu,
                       float* dy, int lddy, float* dgamma, float* dbeta, float* dres, int lddres,
                       int dres_accumulate, void* workspace, size_t workspace_bytes,
                       catseg_stream_t stream);
/* catseg_bn_apply / catseg_bn_backward / catseg_bn_backward_pre / catseg_add_n_act with the output's max |value| folded into
 * amax_record (CATSEG_AMAX_RECORD_BYTES of device memory, zeroed by the caller; NULL = the plain call): the prescale source of
 * catseg_dconv3_f16x2 */
int catseg_bn_apply_amax(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual, int ldr,
                         float* z, int ldz, long long rows, int C, int relu, void* amax_record, catseg_stream_t stream);
int catseg_bn_backward_amax(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* stats,
                            const float* gamma, const float* beta, long long rows, int C, int relu, float* dy, int lddy, float* dgamma,
                            float* dbeta, float* dres, int lddres, int dres_accumulate, void* workspace, size_t workspace_bytes,
                            void* amax_record, catseg_stream_t stream);
/* The ReLU mask of a residual block's output z = relu(bn(y) + residual) as BITS (catseg_bn_mask_bytes(rows, C) = rows C / 8 bytes: bit (e & 7) of byte
 * e >> 3 for the flat element index e = r C + c; C % 8 == 0): catseg_bn_apply_mask / catseg_bn_apply_planes_mask = catseg_bn_apply_amax /
 * catseg_bn_apply_planes (relu on) that also write the mask; catseg_bn_backward_mask / catseg_bn_backward_planes_mask = catseg_bn_backward_amax /
 * catseg_bn_backward_planes reading the bits instead of z in both passes (nn.BatchNorm2d + residual + ReLU of the blocks of models/HRNetv2.py:36-106
 * and of torchvision's BasicBlock / Bottleneck). */
size_t catseg_bn_mask_bytes(long long rows, int C);
int catseg_bn_apply_mask(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual, int ldr, float* z,
                         int ldz, long long rows, int C, void* amax_record, void* mask, catseg_stream_t stream);
int catseg_bn_backward_mask(const float* dz, int lddz, const void* mask, const float* y, int ldy, const float* stats, const float* gamma,
                            long long rows, int C, float* dy, int lddy, float* dgamma, float* dbeta, float* dres, int lddres, int dres_accumulate,
                            void* workspace, size_t workspace_bytes, void* amax_record, catseg_stream_t stream);
int catseg_bn_apply_planes_mask(const float* y, int ldy, const float* mean, const float* scale, const float* beta, const float* residual, int ldr,
                                const void* residual_record, float* z, int ldz, void* z_planes, long long rows, int C, void* z_record, void* mask,
                                catseg_stream_t stream);
int catseg_bn_backward_planes_mask(const float* dz, int lddz, const void* mask, const float* y, int ldy, const float* stats, const float* gamma,
                                   long long rows, int C, void* dy_planes, void* dy_record, void* g_record, const void* y_record, float* dgamma,
                                   float* dbeta, float* dres, int lddres, int dres_accumulate, void* workspace, size_t workspace_bytes,
                                   catseg_stream_t stream);
int catseg_bn_backward_pre_amax(const float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma,
                                const float* partials, int n_blocks, long long rows, int C, float* dq, int lddq, float* dgamma,
                                float* dbeta, void* workspace, size_t workspace_bytes, void* amax_record, catseg_stream_t stream);
int catseg_add_n_act_amax(const float* const* in, const int* ld, int n, float* out, int ldo, long long rows, int C, int relu,
                          void* amax_record, catseg_stream_t stream);
/* the rest of catseg_bn_backward when g (already masked) and the per-block sums [n_blocks][2][C] of g and g * xhat come from
 * catseg_dconv3_bnbwd: merges the sums (dgamma, dbeta) and writes dq = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)).
 * workspace >= 2 * C floats (rounded up to 256 bytes). */
int catseg_bn_backward_pre(const float* g, int ldg, const float* q, int ldq, const float* stats, const float* gamma,
                           const float* partials, int n_blocks, long long rows, int C, float* dq, int lddq, float* dgamma,
                           float* dbeta, void* workspace, size_t workspace_bytes, catseg_stream_t stream);
/* eval-mode / frozen-statistics backward is not on the training path and is not provided. */

/* ---- layout / pointwise helpers ---------------------------------------------------------- */
/* img.float() NCHW (B,3,H,W) -> NHWC with 4 channels (4th = 0) : managers/OCRNet_Manager.py:86 */
int catseg_nchw3_to_nhwc4(const float* x, float* y, int B, int H, int W, catseg_stream_t stream);
/* OIHW-logical/OHWI-physical 7x7x3 stem weight -> packed [O][7][8][4] and back (gradient) */
int catseg_stem_pack_weight(const float* w_ohwi, float* packed, int O, catseg_stream_t stream);
int catseg_stem_unpack_grad(const float* packed_grad, float* dw_ohwi, int O, catseg_stream_t stream);
/* dst[p, c] (+)= alpha * src[p, c] */
int catseg_axpy2d(const float* src, int lds, float* dst, int ldd, long long rows, int C, float alpha,
                  int accumulate, catseg_stream_t stream);
/* out = act(in_0 + ... + in_{n-1}), n <= 4 (HRNet fuse sum, models/HRNetv2.py:237-261); g = dz * (z > 0) */
int catseg_add_n_act(const float* const* in, const int* ld, int n, float* out, int ldo, long long rows, int C,
                     int relu, catseg_stream_t stream);
int catseg_relu_bwd(const float* dz, int lddz, const float* z, int ldz, float* g, int ldg, long long rows, int C,
                    catseg_stream_t stream);
/* conv weight [O][taps][cin] -> zero-padded [O][taps][cpad] (unpad = 0) or back (unpad = 1): lets a
 * 3-channel 3x3 stem (models/HRNetv2.py:311) run on the 4-channel NHWC image */
int catseg_weight_pad_cin(const float* w, float* out, int O, int taps, int cin, int cpad, int unpad,
                          catseg_stream_t stream);
/* x[i] *= s[0], s on the device (applies an upstream loss gradient without a host sync) */
int catseg_scale_by_device_scalar(float* x, long long n, const float* s, catseg_stream_t stream);
/* nn.MaxPool2d(3, 2, 1) of the torchvision stem; idx = window position of the max (uint8) */
int catseg_maxpool3x3s2_fwd(const float* x, int ldx, float* y, int ldy, uint8_t* idx, int B, int H, int W,
                            int C, int Ho, int Wo, catseg_stream_t stream);
int catseg_maxpool3x3s2_bwd(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx, int B,
                            int H, int W, int C, int Ho, int Wo, catseg_stream_t stream);
/* F.interpolate(mode='bilinear') — models/OCR.py:128-131, DeepLabv3Plus.py:68,123,163,
 * HRNetv2.py:253-256,504-512.  NHWC, C channels, input ld ldx, output ld ldy. */
int catseg_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, int Ho,
                        int Wo, int align_corners, int accumulate, catseg_stream_t stream);
/* dx[., c] = sum of dy contributions (deterministic gather form); columns [C, zero_to) of dx zeroed.
 * workspace >= B*H*Wo*C*4 bytes. */
int catseg_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int C,
                        int Ho, int Wo, int align_corners, int zero_to, int accumulate, void* workspace,
                        size_t workspace_bytes, catseg_stream_t stream);
/* nn.AdaptiveAvgPool2d(1) (ASPP image pooling, DeepLabv3Plus.py:121) and its backward */
int catseg_global_avgpool_fwd(const float* x, int ldx, float* y, int B, int HW, int C,
                              catseg_stream_t stream);
int catseg_global_avgpool_bwd(const float* dy, float* dx, int lddx, int B, int HW, int C, int accumulate,
                              catseg_stream_t stream);

/* nn.AdaptiveAvgPool2d(S) (UPerNet pyramid pooling, models/UPerNet.py:25-33); y is [B][S][S][C] compact */
int catseg_adaptive_avgpool_fwd(const float* x, int ldx, float* y, int B, int H, int W, int C, int S,
                                catseg_stream_t stream);
int catseg_adaptive_avgpool_bwd(const float* dy, float* dx, int lddx, int B, int H, int W, int C, int S,
                                int accumulate, catseg_stream_t stream);

/* ---- softmax ------------------------------------------------------------------------------ */
/* F.softmax(probs.view(B,K,N), dim=2) at models/OCR.py:165, on NHWC logits [B][N][ld]:
 * per (b, k) over the N pixels, K <= 32.  Columns [K, ld) of the output are zeroed. */
size_t catseg_softmax_spatial_workspace(int B, int N);
int catseg_softmax_spatial_fwd(const float* x, float* y, int B, int N, int K, int ld, void* workspace,
                               size_t workspace_bytes, catseg_stream_t stream);
int catseg_softmax_spatial_bwd(const float* y, const float* dy, float* dx, int B, int N, int K, int ld,
                               int accumulate, void* workspace, size_t workspace_bytes, catseg_stream_t stream);
/* F.softmax(scale * sim, dim=-1) at models/OCR.py:270-271: per row over K (<= 64) columns */
int catseg_softmax_rows_fwd(const float* x, float* y, long long rows, int K, int ld, float scale,
                            catseg_stream_t stream);
int catseg_softmax_rows_bwd(const float* y, const float* dy, float* dx, long long rows, int K, int ld,
                            float scale, catseg_stream_t stream);

/* ---- losses ------------------------------------------------------------------------------- */
/* LovaszSoftmax.forward (losses/LovaszSoftmax.py:19-61, per_image=False, classes 'present',
 * nothing ignored) fused with its autograd backward.  logits [P][K] (ld = K, compact),
 * labels int64 [P].  loss_out[0] = loss; dlogits (may be NULL) = weight * dloss/dlogits. */
size_t catseg_lovasz_workspace(long long P, int K);
int catseg_lovasz_softmax(const float* logits, const int64_t* labels, long long P, int K, float weight,
                          float* loss_out, float* dlogits, int accumulate_dlogits, void* workspace,
                          size_t workspace_bytes, catseg_stream_t stream);
/* the same pipeline in two calls around autograd (losses/LovaszSoftmax.py:19-32 + its backward): _fwd leaves d loss / d prob and the class
   counts in `workspace` (catseg_lovasz_workspace bytes, kept untouched by the caller together with `logits`) when want_grad != 0; _bwd writes
   (d loss / d logit) * upstream, upstream = a DEVICE scalar (autograd's grad_output; null = 1): no separate scaling pass over the gradient */
int catseg_lovasz_softmax_fwd(const float* logits, const int64_t* labels, long long P, int K, float weight, float* loss_out,
                              int want_grad, void* workspace, size_t workspace_bytes, catseg_stream_t stream);
int catseg_lovasz_softmax_bwd(const float* logits, long long P, int K, float weight, const float* upstream, float* dlogits,
                              int accumulate_dlogits, void* workspace, size_t workspace_bytes, catseg_stream_t stream);
/* nn.CrossEntropyLoss(ignore_index) (losses/LossWrapper.py:17-24) fused with backward */
size_t catseg_ce_workspace(long long P);
int catseg_cross_entropy(const float* logits, const int64_t* labels, long long P, int K, long long ignore_index,
                         float weight, float* loss_out, float* dlogits, void* workspace,
                         size_t workspace_bytes, catseg_stream_t stream);
/* OhemCrossEntropy.forward (losses/OhemCrossEntropy.py:22-39) fused with backward: mean CE over the non-ignored
 * pixels whose target-class probability is < max(thresh, k-th smallest probability), k = min(min_kept, n - 1).
 * The k-th order statistic is found by a 3-pass radix select on device (no sort, no host sync). */
size_t catseg_ohem_workspace(long long P);
int catseg_ohem_cross_entropy(const float* logits, const int64_t* labels, long long P, int K, long long ignore_index,
                              float thresh, long long min_kept, float weight, float* loss_out, float* dlogits,
                              void* workspace, size_t workspace_bytes, catseg_stream_t stream);

/* ---- input side -------------------------------------------------------------- */
/* Dataset_from_df.__getitem__ + its deterministic transforms (datasets/Dataset_from_df.py:31-69;
 * remap_mask utils/utils.py:23-47; FlipNP / PadNP utils/transforms.py:222-240, 8-20; ToTensor / Normalize
 * utils/utils.py:440-447) for a whole batch on device.
 *   img u8 [B][H][W][3] RGB, lbl u8 [B][H][W] raw ids; lut u8[256] (NULL = identity);
 *   flips int32 [B] (NULL = none): bit 0 = horizontal, bit 1 = vertical, applied before the padding;
 *   rows are reflect-padded (np.pad 'reflect') by pad_top / pad_bottom; x = u8 / 255, then (x - mean) / std if given.
 *   Outputs: x_nchw f32 [B][3][H'][W] and / or x_nhwc4 f32 [B][H'][W][4] (4th channel 0: the stem layout),
 *   labels int64 [B][H'][W], H' = H + pad_top + pad_bottom.  Either side (image / label) may be NULL. */
int catseg_ingest_u8(const uint8_t* img, const uint8_t* lbl, int B, int H, int W, const uint8_t* lut, const int32_t* flips,
                     int pad_top, int pad_bottom, const float* mean, const float* stdv, float* x_nchw, float* x_nhwc4,
                     int64_t* labels, catseg_stream_t stream);

/* PIL-based training augmentations on uint8 NHWC batches [B][H][W][3], bit-exact with Pillow (csrc/augment.hip; reference call
 * sites utils/utils.py:412-417, utils/transforms.py:242-251):
 *   catseg_aug_pad_flip_u8: FlipNP (bit 0 horizontal, bit 1 vertical) + PadNP reflect rows, uint8 -> uint8 (what precedes ToPILImage);
 *   catseg_aug_box_blur: ONE extended box-blur pass along axis (0 = H, 1 = W) with per-image (radius, ww, fw) of BoxBlur.c
 *     (radius[b] < 0: copy); ImageFilter.GaussianBlur(r) = 3 passes along W, then 3 along H, box radius from r (host side);
 *   catseg_aug_color_op: ONE operation of torchvision ColorJitter per image, in place: op[b] = 0 brightness, 1 contrast,
 *     2 saturation (factor[b] = the enhance factor), 3 hue (factor[b] = the integer H shift 0..255), < 0 none; workspace >= 8 B bytes. */
int catseg_aug_pad_flip_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, int C, const int32_t* flips, int pad_top,
                           int pad_bottom, catseg_stream_t stream);
int catseg_aug_box_blur(const uint8_t* in, uint8_t* out, int B, int H, int W, int axis, const int32_t* radius, const uint32_t* ww,
                        const uint32_t* fw, catseg_stream_t stream);
int catseg_aug_color_op(uint8_t* img, int B, int H, int W, const int32_t* op, const float* factor, void* workspace,
                        size_t workspace_bytes, catseg_stream_t stream);

/* Test-time augmentation plumbing (managers/BaseManager.py:652-660 wraps the model in ttach
 * HorizontalFlip x Scale(0.75, 1, 1.5, 1.75, 2), merge 'mean'; ttach is an un-vendored dependency: its Scale is
 * F.interpolate(mode='nearest', size=(int(h*s), int(w*s)))).  NHWC nearest resize:
 *   dst[b][y][x][:] (+)= src[b][sy][sx][:], sy = min(floor(y * (float)Hi / Ho), Hi - 1), sx likewise;
 *   flip 1: the source is flipped horizontally first (image augmentation), flip 2: the result is flipped
 *   (mask de-augmentation); accumulate: add into dst; divide_by != 0: the stored value is divided by it
 *   (the 'mean' merge on the last accumulation). */
int catseg_resize_nearest(const float* src, int lds, float* dst, int ldd, int B, int Hi, int Wi, int Ho, int Wo, int C, int flip,
                          int accumulate, float divide_by, catseg_stream_t stream);

/* ---- metrics / optimiser ------------------------------------------------------------------ */
/* t_get_confusion_matrix (utils/torch_utils.py:221-241): cm[pred*K + gt] += 1 (int32, K x K),
 * labels >= K are dropped; cm is accumulated into (zero it first for a fresh matrix). */
int catseg_confusion_matrix(const float* logits, const int64_t* labels, long long P, int K, int32_t* cm,
                            catseg_stream_t stream);
/* torch.optim.Adam(lr) step over a flat parameter buffer (managers/BaseManager.py:441) */
int catseg_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                     float beta2, float eps, int step, float grad_scale, catseg_stream_t stream);
/* the same step with its step-dependent scalars in DEVICE memory -- hyper = {lr, 1 - beta1^step, sqrt(1 - beta2^step), grad_scale}, the
   values catseg_adam_hyper (host, no GPU call) computes exactly as catseg_adam_step does -- so that a launch captured in a hipGraph
   (the whole training step of managers/OCRNet_Manager.py:80-90 replayed as one graph) follows the learning-rate schedule and the bias
   correction of torch.optim.Adam from step to step; bit-identical to catseg_adam_step for equal values */
void catseg_adam_hyper(float lr, float beta1, float beta2, int step, float grad_scale, float* hyper4);
int catseg_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1,
                         float beta2, float eps, catseg_stream_t stream);

/* ---- pointwise (1 x 1, stride 1) convolutions in split precision with the split in registers (csrc/pconv1.hip) ----
   Replaces the ATen 1 x 1 conv2d calls (forward, and their autograd backward) of the stage-1 bottlenecks (models/HRNetv2.py:68-106), of the
   object-attention block (models/OCR.py:186-235: f_pixel / f_up) and of the HRNet fuse layers (models/HRNetv2.py:237-261) -- layers too
   small for the blocked-plane kernels above, HBM-bound GEMMs with K = 64 ... 512 -- and, since round 6, of the 1 x 1 layers of torchvision's
   ResNet bottlenecks up to 2048 output columns (models/OCR.py:58-61: 64 <-> 256, 128 <-> 512, 256 <-> 1024 at stride 8; catseg_pconv1_supported:
   32 <= N <= 2048, 32 <= K <= 8160, K % 8 == 0).  The activation operand is the fp32 NHWC tensor itself
   plus the amax record its producer left (CATSEG_AMAX_RECORD_BYTES); the weight operand is a pre-split image.
   catseg_pconv1_prep_batch: entries = DEVICE array of n 64-byte records {int64 weight offset (floats, relative to flat), int64 image offset
   (bytes, relative to wimg_base), int32 O, I, kh, kw, transposed, ky0, kys, nky, kx0, kxs, nkx, pad} (a 1 x 1 layer: kh = kw = nky = nkx =
   kys = kxs = 1, ky0 = kx0 = 0); records = n x {uint32 bits of max|w|, int32 exponent} (DEVICE).
   catseg_pconv1: y[M][N] (+)= x[M][K] . B^T (+ bias); B = image of (N, K) = the layer's [O][I] weights (forward: N = O, K = I) or, with
   transposed != 0 at preparation, their transpose (backward-data: N = I, K = O).  bn_part as catseg_conv2d_fwd_bnstats.
   catseg_pconv1_wgrad: dw[Cout][Cin] = dy^T . x over P pixels (slabs + fixed-order sum: deterministic). */
int catseg_pconv1_supported(int N, int K);
size_t catseg_pconv1_wimg_bytes(int N, int K);
int catseg_pconv1_prep_batch(const float* flat, int n, const void* entries, void* wimg_base, void* records, catseg_stream_t stream);
int catseg_pconv1(long long M, int N, int K, const float* x, int ldx, const void* x_rec, const void* wimg, const void* w_rec,
                  const float* bias, float* y, int ldy, int accumulate, float* bn_part, size_t bn_part_floats, int* tile_rows,
                  int* n_tiles, catseg_stream_t stream);
int catseg_pconv1_wgrad_supported(int Cout, int Cin);
size_t catseg_pconv1_wgrad_workspace(long long P, int Cout, int Cin);
int catseg_pconv1_wgrad(long long P, int Cout, int Cin, const float* dy, int lddy, const void* dy_rec, const float* x, int ldx,
                        const void* x_rec, float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream);
/* The same kernels as "gather" launches for dense convolutions with kh x kw taps, stride 1 / 2, padding, dilation (stride 1) whose input
   carries an amax record: the 3 x 3 / stride 2 layers of the HRNet fuse chains and transitions (models/HRNetv2.py:176-198,237-261: ATen
   conv2d + its autograd backward), the 256 -> 48 transition, the stem's second convolution.  Weight images through
   catseg_pconv1_prep_batch with the 64-byte entries catseg_gconv_entries fills (host memory; 1 entry forward, stride^2 entries -- one
   per input-pixel parity class, catseg_gconv_class_bytes apart -- backward-data).  Results as the fp32 kernels' to split-precision accuracy. */
int catseg_gconv_supported(const catseg_conv_desc* d);
size_t catseg_gconv_class_bytes(const catseg_conv_desc* d);
size_t catseg_gconv_wimg_bytes(const catseg_conv_desc* d, int backward_data);
int catseg_gconv_entries(const catseg_conv_desc* d, int backward_data, long long w_off, long long img_off, void* entries_out);
int catseg_gconv_fwd(const catseg_conv_desc* d, const float* x, const void* x_rec, const void* wimg, const void* w_rec,
                     const float* bias, float* y, float* bn_part, size_t bn_part_floats, int* tile_rows, int* n_tiles,
                     catseg_stream_t stream);
int catseg_gconv_bwd_data(const catseg_conv_desc* d, const float* dy, const void* dy_rec, const void* wimg_classes, const void* w_rec,
                          float* dx, int accumulate, catseg_stream_t stream);
int catseg_gconv_wgrad_supported(const catseg_conv_desc* d);
size_t catseg_gconv_wgrad_workspace(const catseg_conv_desc* d);
int catseg_gconv_bwd_weight(const catseg_conv_desc* d, const float* dy, const void* dy_rec, const float* x, const void* x_rec,
                            float* dw, void* workspace, size_t workspace_bytes, catseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CATSEG_H */
