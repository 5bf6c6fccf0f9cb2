"""Micro-benchmark of the implicit-GEMM kernels on the OCRNet-R50 layer shapes (bs 8, 544x960)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops

dev = torch.device("cuda")
SHAPES = [  # name, B,H,W,Cin,Cout,k,s,p,d
    ("l1.1x1 64>256", 8, 136, 240, 64, 256, 1, 1, 0, 1),
    ("l1.3x3 64>64", 8, 136, 240, 64, 64, 3, 1, 1, 1),
    ("l1.1x1 256>64", 8, 136, 240, 256, 64, 1, 1, 0, 1),
    ("l2.3x3 128>128", 8, 68, 120, 128, 128, 3, 1, 1, 1),
    ("l2.1x1 128>512", 8, 68, 120, 128, 512, 1, 1, 0, 1),
    ("l3.1x1 1024>256", 8, 68, 120, 1024, 256, 1, 1, 0, 1),
    ("l3.3x3d2 256>256", 8, 68, 120, 256, 256, 3, 1, 2, 2),
    ("l3.1x1 256>1024", 8, 68, 120, 256, 1024, 1, 1, 0, 1),
    ("l4.1x1 2048>512", 8, 68, 120, 2048, 512, 1, 1, 0, 1),
    ("l4.3x3d4 512>512", 8, 68, 120, 512, 512, 3, 1, 4, 4),
    ("l4.1x1 512>2048", 8, 68, 120, 512, 2048, 1, 1, 0, 1),
    ("high_map 3x3 2048>512", 8, 68, 120, 2048, 512, 3, 1, 1, 1),
    ("interm 3x3 1024>512", 8, 68, 120, 1024, 512, 3, 1, 1, 1),
]

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

rows = []
for name, B, H, W, Ci, Co, k, s, p, d in SHAPES:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, Co, k, k, s, p, d)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * y.numel() * Ci * k * k
    t_f = timeit(lambda: ops.conv_fwd(x, w, None, Co, k, k, s, p, d, out=y))
    t_d = timeit(lambda: ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, s, p, d, out=dx))
    t_w = timeit(lambda: ops.conv_bwd_weight(x, dy, dw, None, k, k, s, p, d))
    rows.append((name, fl / 1e9, t_f, fl / t_f / 1e9, t_d, fl / t_d / 1e9, t_w, fl / t_w / 1e9))
    print("%-24s %7.1f GF | fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF" % rows[-1], flush=True)
tot = [sum(r[i] for r in rows) for i in (1, 2, 4, 6)]
print("sum GF %.1f  fwd %.2f ms (%.1f TF)  dgrad %.2f ms (%.1f TF)  wgrad %.2f ms (%.1f TF)" % (
    tot[0], tot[1], tot[0] / tot[1], tot[2], tot[0] / tot[2], tot[3], tot[0] / tot[3]))
