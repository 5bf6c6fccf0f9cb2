"""Run the direct 3x3 kernel alone (for rocprofv3 passes).  usage: run_dconv3.py [fwd|dgrad|h2fwd|h2dgrad] [n] [B,H,W,C] [blocks]
(h2*: the two-plane fp16 build; eight inputs in turn, so that they are not cache resident)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd._lib import lib
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, H, W, C = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8,136,240,48").split(",")]
if len(sys.argv) > 4:
    lib.catseg_debug_set_dconv3_blocks(int(sys.argv[4]))
dev = torch.device("cuda")
x = torch.randn(B, H, W, C, device=dev)
w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
y = torch.empty_like(x)
if kind == "plwgrad":            # backward-weight on planes (csrc/dwgrad3_pl.hip), eight tensor pairs in turn
    xs = [torch.randn(B, H, W, C, device="cuda") for _ in range(8)]
    dys = [torch.randn(B, H, W, C, device="cuda") * 1e-3 for _ in range(8)]
    xps = [ops.planes_from_f32(x) for x in xs]
    dps = [ops.planes_from_f32(d) for d in dys]
    dw = torch.empty(C, C, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    for i in range(n):
        ops.dwgrad3_pl(xps[i % 8], dps[i % 8], dw)
elif kind.startswith("pl"):       # the planes kernel (csrc/dconv3_pl.hip): producer-written planes of eight inputs in turn
    xs = [torch.randn(B, H, W, C, device=dev) for _ in range(8)]
    xps = [ops.planes_from_f32(t) for t in xs]
    wimg = ops.dconv3_weight_image(w, backward_data=(kind == "pldgrad"), h2=True)
    torch.cuda.synchronize()
    for i in range(n):
        ops.dconv3_pl(xps[i % 8], wimg, None, out=y, bn_stats=(kind == "plfwd"))
elif kind.startswith("h2"):
    xs = [torch.randn(B, H, W, C, device=dev) for _ in range(8)]
    for t in xs:
        t._amax = ops.new_amax(dev)
        t._amax[0:1] = t.abs().max().reshape(1).view(torch.int32)
    wimg = ops.dconv3_weight_image(w, backward_data=(kind == "h2dgrad"), h2=True)
    torch.cuda.synchronize()
    for i in range(n):
        ops.dconv3(xs[i % 8], wimg, None, out=y, bn_stats=(kind == "h2fwd"), x_amax=xs[i % 8]._amax)
else:
    wimg = ops.dconv3_weight_image(w, backward_data=(kind == "dgrad"))
    torch.cuda.synchronize()
    for _ in range(n):
        ops.dconv3(x, wimg, None, out=y, bn_stats=(kind == "fwd"))
torch.cuda.synchronize()
print("done", kind, 2.0 * B * H * W * C * C * 9 / 1e9, "GF per launch")
