"""Run the direct 3x3 kernel alone (for rocprofv3 passes).  usage: run_dconv3.py [fwd|dgrad] [n] [B,H,W,C] [blocks]"""
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
wimg = ops.dconv3_weight_image(w, backward_data=(kind == "dgrad"))
torch.cuda.synchronize()
for _ in range(n):
    ops.dconv3(x, wimg, None, out=y, bn_stats=(kind == "fwd"))
torch.cuda.synchronize()
print("done", kind, 2.0 * B * H * W * C * C * 9 / 1e9, "GF per launch")
