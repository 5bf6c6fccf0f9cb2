"""the exact-fp32 implicit-GEMM kernels (csrc/igemm.hip) on the layer shapes that still take them in an HRNet-W48 step: forward with BatchNorm
partials / backward-data, microseconds per launch (four tensors in turn).   python3 tools/time_f32.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
SH = [((8, 136, 240, 64), 256, 1, 1, 0, "layer1 64->256 1x1"), ((8, 136, 240, 256), 64, 1, 1, 0, "layer1 256->64 1x1"), ((8, 136, 240, 64), 64, 3, 1, 1, "layer1 64->64 3x3"),
      ((8, 272, 480, 64), 64, 3, 2, 1, "stem conv2 3x3/2"), ((8, 136, 240, 256), 48, 3, 1, 1, "transition 256->48"), ((8, 136, 240, 512), 25, 1, 1, 0, "class head 512->25"),
      ((8, 68, 120, 96), 192, 3, 2, 1, "fuse 96->192 3x3/2"), ((8, 34, 60, 192), 96, 1, 1, 0, "fuse 192->96 1x1")]


def timeit(fn, n=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for xs, Cout, k, s, pd, name in SH:
    B, H, W, Cin = xs
    x = [torch.randn(B, H, W, Cin, device=dev).relu_() for _ in range(4)]
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    Ho, Wo = ops.conv_out_size(H, k, s, pd, 1), ops.conv_out_size(W, k, s, pd, 1)
    ld = max(32, (Cout + 3) // 4 * 4) if Cout % 4 else Cout
    dy = [(torch.randn(B, Ho, Wo, ld, device=dev) * 1e-4)[..., :Cout] for _ in range(4)]
    tf = timeit(lambda i: ops.conv_fwd(x[i % 4], w, None, Cout, k, k, s, pd, 1, bn_stats=True, exact=True))
    td = timeit(lambda i: ops.conv_bwd_data(dy[i % 4], w, xs, k, k, s, pd, 1))
    dw = torch.empty_like(w)
    tw = timeit(lambda i: ops.conv_bwd_weight(x[i % 4], dy[i % 4], dw, None, k, k, s, pd, 1))
    print("%-22s fwd %7.1f us   dgrad %7.1f us   wgrad %7.1f us" % (name, tf, td, tw), flush=True)
    ops.release_b3_cache()
