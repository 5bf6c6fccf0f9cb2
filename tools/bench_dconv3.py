"""Direct split-precision 3x3 convolution (csrc/dconv3_b3.hip) against the fp32 implicit GEMM on the HRNet-W48 trunk shapes.
usage: bench_dconv3.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd._lib import lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
SHAPES = [(8, 136, 240, 48), (8, 68, 120, 96), (8, 34, 60, 192), (8, 17, 30, 384), (8, 136, 240, 64)]


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


for (B, H, W, C) in SHAPES:
    x = torch.randn(B, H, W, C, device=dev)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x)
    gf = 2.0 * B * H * W * C * C * 9 / 1e9
    ops.PRECISION = "fp32"
    t_f32 = timed(lambda: ops.conv_fwd(x, w, None, C, 3, 3, 1, 1, 1, out=y, bn_stats=True), reps)
    t_f32d = timed(lambda: ops.conv_bwd_data(x, w, tuple(x.shape), 3, 3, 1, 1, 1, out=y), reps)
    dw = torch.empty_like(w)
    dyt = torch.randn_like(x)
    t_f32w = timed(lambda: ops.conv_bwd_weight(x, dyt, dw, None, 3, 3, 1, 1, 1), reps)
    line = "C=%3d %dx%d  fp32 igemm fwd %7.1f us %6.1f TF  dgrad %7.1f us %6.1f TF  wgrad %7.1f us %6.1f TF" % (
        C, H, W, t_f32, gf / t_f32 * 1e3, t_f32d, gf / t_f32d * 1e3, t_f32w, gf / t_f32w * 1e3)
    if lib.catseg_dwgrad3_supported(C):
        for blocks in (512, 256, 1024):
            lib.catseg_debug_set_dwgrad3_blocks(blocks)
            t_w = timed(lambda: ops.dwgrad3(x, dyt, dw), reps)
            line += "  | d3 wgrad[%d] %7.1f us %6.1f TF" % (blocks, t_w, gf / t_w * 1e3)
        lib.catseg_debug_set_dwgrad3_blocks(0)
    if lib.catseg_dconv3_supported(C):
        wimg = ops.dconv3_weight_image(w)
        wimg_t = ops.dconv3_weight_image(w, backward_data=True)
        t_d = timed(lambda: ops.dconv3(x, wimg, None, out=y, bn_stats=True), reps)
        t_dd = timed(lambda: ops.dconv3(x, wimg_t, None, out=y), reps)
        lib.catseg_debug_set_dconv3_spec(1)
        t_s = timed(lambda: ops.dconv3(x, wimg, None, out=y, bn_stats=True), reps)
        t_sd = timed(lambda: ops.dconv3(x, wimg_t, None, out=y), reps)
        lib.catseg_debug_set_dconv3_spec(0)
        t_d = timed(lambda: ops.dconv3(x, wimg, None, out=y, bn_stats=True), reps)
        t_dd = timed(lambda: ops.dconv3(x, wimg_t, None, out=y), reps)
        lib.catseg_debug_set_dconv3_spec(-1)
        line += "  | SPEC fwd %7.1f us %6.1f TF  dgrad %7.1f us %6.1f TF" % (t_s, gf / t_s * 1e3, t_sd, gf / t_sd * 1e3)
        if C == 96:
            lib.catseg_debug_set_dconv3_alt96(1)
            t_a = timed(lambda: ops.dconv3(x, wimg, None, out=y, bn_stats=True), reps)
            t_ad = timed(lambda: ops.dconv3(x, wimg_t, None, out=y), reps)
            lib.catseg_debug_set_dconv3_alt96(0)
            line += "  | 4x32-uniform fwd %7.1f us dgrad %7.1f us" % (t_a, t_ad)
        t_prep = timed(lambda: lib.catseg_dconv3_prep(w.data_ptr(), C, 0, wimg.data_ptr(), torch.cuda.current_stream().cuda_stream), reps)
        line += "  | direct fwd %7.1f us %6.1f TF  dgrad %7.1f us %6.1f TF  prep %5.1f us" % (t_d, gf / t_d * 1e3, t_dd, gf / t_dd * 1e3, t_prep)
    print(line, flush=True)
    ops.release_b3_cache()
