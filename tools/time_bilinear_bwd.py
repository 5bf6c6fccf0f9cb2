"""standalone times of catseg_bilinear_bwd at the bench's shapes: one launch with the row in LDS (fused) against the two separable passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
CASES = [("logits 25ch 136x240 <- 544x960", 8, 136, 240, 25, 544, 960, 25, 32),
         ("concat 384ch 17x30 <- 136x240 (ld 720)", 8, 17, 30, 384, 136, 240, 720, 0),
         ("concat 192ch 34x60 <- 136x240 (ld 720)", 8, 34, 60, 192, 136, 240, 720, 0),
         ("concat 96ch 68x120 <- 136x240 (ld 720)", 8, 68, 120, 96, 136, 240, 720, 0),
         ("fuse 96ch 68x120 -> 48? 136x240 (ld 48)", 8, 68, 120, 48, 136, 240, 48, 0),
         ("fuse 384->192 17x30 <- 34x60", 8, 17, 30, 192, 34, 60, 192, 0)]
for name, B, H, W, C, Ho, Wo, ld, zt in CASES:
    bufs = [torch.randn(B, Ho, Wo, ld, device=dev) for _ in range(3)]       # rotate: not cache-resident
    dys = [b[..., :C] if ld != C else b for b in bufs]
    res = {}
    for fused in (1, 0, 1, 0):
        ops.lib.catseg_debug_set_bilinear_bwd_fused(fused)
        for i in range(3):
            ops.bilinear_bwd(dys[i % 3], (B, H, W, C), False, zero_to=zt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 12
        for i in range(n):
            ops.bilinear_bwd(dys[i % 3], (B, H, W, C), False, zero_to=zt)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(fused, []).append(e0.elapsed_time(e1) / n * 1e3)
    gb = 4.0 * (B * Ho * Wo * C + B * H * W * C) / 1e9
    print("%-44s fused %7.1f us (%.2f TB/s)   two-pass %7.1f us (%.2f TB/s)" % (name, min(res[1]), gb / min(res[1]) * 1e3, min(res[0]), gb / min(res[0]) * 1e3), flush=True)
ops.lib.catseg_debug_set_bilinear_bwd_fused(1)
