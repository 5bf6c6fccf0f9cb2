"""Compare igemm block tiles on a few layer shapes (forced through catseg_debug_set_tile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops, _lib

dev = torch.device("cuda")
SHAPES = [
    ("l1.3x3 64>64", 8, 136, 240, 64, 64, 3, 1, 1, 1),
    ("l1.1x1 64>256", 8, 136, 240, 64, 256, 1, 1, 0, 1),
    ("l2.3x3 128>128", 8, 68, 120, 128, 128, 3, 1, 1, 1),
    ("l3.1x1 1024>256", 8, 68, 120, 1024, 256, 1, 1, 0, 1),
    ("l3.3x3d2 256>256", 8, 68, 120, 256, 256, 3, 1, 2, 2),
    ("l4.1x1 2048>512", 8, 68, 120, 2048, 512, 1, 1, 0, 1),
    ("l4.3x3d4 512>512", 8, 68, 120, 512, 512, 3, 1, 4, 4),
    ("l4.1x1 512>2048", 8, 68, 120, 512, 2048, 1, 1, 0, 1),
    ("interm 3x3 1024>512", 8, 68, 120, 1024, 512, 3, 1, 1, 1),
]
TILES = [(0, 0), (2, 2), (4, 2), (2, 4), (4, 4), (2, 1), (1, 2), (1, 1)]

def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for name, B, H, W, Ci, Co, k, s, p, d in SHAPES:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, Co, k, k, s, p, d)
    dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * y.numel() * Ci * k * k
    line = "%-22s" % name
    for (mi, ni) in TILES:
        _lib.lib.catseg_debug_set_tile(mi, ni)
        tf = timeit(lambda: ops.conv_fwd(x, w, None, Co, k, k, s, p, d, out=y))
        td = timeit(lambda: ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, s, p, d, out=dx))
        tw = timeit(lambda: ops.conv_bwd_weight(x, dy, dw, None, k, k, s, p, d))
        line += " | %dx%d f%5.1f d%5.1f w%5.1f" % (mi, ni, fl / tf / 1e9, fl / td / 1e9, fl / tw / 1e9)
    _lib.lib.catseg_debug_set_tile(0, 0)
    print(line, flush=True)
