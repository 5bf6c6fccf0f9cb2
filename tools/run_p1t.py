"""Run the pointwise backward-weight kernel (csrc/pconv1.hip: p1t_kernel) alone for rocprofv3 passes.  usage: run_p1t.py [n] [rows,K,N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rows, K, N = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "261120,512,256").split(",")]
dev = torch.device("cuda")
xs = [torch.randn(rows, K, device=dev).relu_() for _ in range(4)]
dys = [torch.randn(rows, N, device=dev) * 1e-5 for _ in range(4)]
for t in xs + dys:
    t._amax = ops.new_amax(dev)
    t._amax[0] = t.abs().max().reshape(1).view(torch.int32)[0]
dw = torch.empty(N, K, device=dev)
need = ops.lib.catseg_pconv1_wgrad_workspace(rows, N, K)
ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for i in range(n):
    ops.check(ops.lib.catseg_pconv1_wgrad(rows, N, K, ops.ptr(dys[i % 4]), N, ops.ptr(dys[i % 4]._amax), ops.ptr(xs[i % 4]), K, ops.ptr(xs[i % 4]._amax),
                                          ops.ptr(dw), ops.ptr(ws), need, ops.stream()))
torch.cuda.synchronize()
