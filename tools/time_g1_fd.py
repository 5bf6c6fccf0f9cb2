"""forward / backward-data of the gather route (csrc/pconv1.hip: p1_kernel<..., GEO>) on the strided 3 x 3 layers of an HRNet-W48 step,
microseconds per launch (four tensors in turn); CATSEG_LIB selects the library for A/B runs.   python3 tools/time_g1_fd.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")


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


for xs, Cout, k, s, pd, name in [((8, 136, 240, 48), 96, 3, 2, 1, "48->96 3x3/2"), ((8, 136, 240, 48), 48, 3, 2, 1, "48->48 3x3/2"), ((8, 68, 120, 96), 192, 3, 2, 1, "96->192 3x3/2"),
                                 ((8, 68, 120, 96), 96, 3, 2, 1, "96->96 3x3/2"), ((8, 136, 240, 256), 96, 3, 2, 1, "256->96 3x3/2"), ((8, 136, 240, 256), 48, 3, 1, 1, "256->48 3x3")]:
    B, H, W, Cin = xs
    x = [torch.randn(B, H, W, Cin, device=dev).relu_() for _ in range(4)]
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    Ho, Wo = ops.conv_out_size(H, k, s, pd, 1), ops.conv_out_size(W, k, s, pd, 1)
    dy = [torch.randn(B, Ho, Wo, Cout, device=dev) * 1e-4 for _ in range(4)]
    for t in x + dy:
        t._amax = ops.new_amax(dev)
        t._amax[0] = t.abs().max().reshape(1).view(torch.int32)[0]
    dx = torch.zeros(B, H, W, Cin, device=dev)
    ops.PROFILE = []
    tf = timeit(lambda i: ops.conv_fwd(x[i % 4], w, None, Cout, k, k, s, pd, 1, bn_stats=True))
    td = timeit(lambda i: ops.conv_bwd_data(dy[i % 4], w, xs, k, k, s, pd, 1, out=dx, accumulate=True))
    kinds = sorted(set(q[0] for q in ops.PROFILE))
    ops.PROFILE = None
    print("%-16s fwd %7.1f us  dgrad (accumulating) %7.1f us  %s" % (name, tf, td, kinds), flush=True)
    ops.release_b3_cache()
