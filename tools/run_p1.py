"""Run the pointwise kernel (csrc/pconv1.hip) alone for rocprofv3 passes.  usage: run_p1.py [fwd|dgrad] [n] [rows,K,N]   (four inputs in turn)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows, K, N = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "261120,512,256").split(",")]
dev = torch.device("cuda")
xs = [torch.randn(1, 1, rows, K, device=dev).relu_() for _ in range(4)]
for t in xs:
    t._amax = ops.new_amax(dev)
    t._amax[0] = t.abs().max().reshape(1).view(torch.int32)[0]
w = (torch.randn(N, K, 1, 1, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
outs = [torch.empty(1, 1, rows, N, device=dev) for _ in range(2)]
wimg = ops.p1_weight_image(w)
torch.cuda.synchronize()
for i in range(n):
    ops.pconv1(xs[i % 4], wimg, None, N, outs[i % 2], bn_stats=(kind == "fwd"))
torch.cuda.synchronize()
