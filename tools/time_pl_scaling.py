"""Does a direct trunk launch pay a fixed cost?  The planes kernel (catseg_dconv3_pl) and the in-kernel-split kernel (catseg_dconv3_f16x2) on
B = 4, 8, 16, 32 frames of one branch shape: time = a + b * tiles separates the per-launch cost a (dispatch, block start-up on 80 KB of LDS,
prologue, tail, end-of-kernel release) from the per-tile cost b.  Usage: python tools/time_pl_scaling.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def t(fn, n):
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(n):
            fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3


for (H, W, C) in [(68, 120, 96), (136, 240, 48), (34, 60, 192), (17, 30, 384)]:
    res = []
    for B in (2, 4, 8, 16, 32):
        n = 4
        xs = [torch.randn(B, H, W, C, device=dev) for _ in range(n)]
        for x in xs:
            x._amax = ops.new_amax(dev)
            x._amax[0:1] = x.abs().max().reshape(1).view(torch.int32)
        w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        wimg = ops.dconv3_weight_image(w, h2=True)
        xps = [ops.planes_from_f32(x, x._amax) for x in xs]
        y = torch.empty_like(xs[0])
        old = t(lambda i: ops.dconv3(xs[i], wimg, None, out=y, bn_stats=True, x_amax=xs[i]._amax), n)
        new = t(lambda i: ops.dconv3_pl(xps[i], wimg, None, out=y, bn_stats=True), n)
        dys = [torch.randn(B, H, W, C, device=dev) * 1e-3 for _ in range(n)]
        for d in dys:
            d._amax = ops.new_amax(dev)
            d._amax[0:1] = d.abs().max().reshape(1).view(torch.int32)
        wnew = float("nan")
        if ops.lib.catseg_dwgrad3_pl_supported(C):
            dps = [ops.planes_from_f32(d, d._amax) for d in dys]
            dw = torch.empty_like(w)
            wnew = t(lambda i: ops.dwgrad3_pl(xps[i], dps[i], dw), n)
        gf = 2.0 * B * H * W * C * C * 9 / 1e9
        res.append((B, old, new, wnew, gf))
        print("C=%3d B=%2d %dx%d: split %.1f us, planes %.1f us (%.2f of 833), wgrad planes+reduce %.1f us (%.2f)" % (C, B, H, W, old, new, gf / new / 833.3, wnew, gf / wnew / 833.3), flush=True)
        del xs, xps, dys
        ops.release_b3_cache()
    (b0, o0, n0, w0, _), (b1, o1, n1, w1, _) = res[2], res[4]
    for name, v0, v1 in (("split", o0, o1), ("planes", n0, n1), ("wgrad", w0, w1)):
        b = (v1 - v0) / (b1 - b0)
        print("   %-7s per frame %.2f us, fixed %.1f us (from B = %d and %d)" % (name, b, v0 - b * b0, b0, b1))
