"""direct backward-weight kernel: target block count sweep (slab traffic vs occupancy)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops, _lib
dev = torch.device("cuda")
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for nm, B, H, W, C in [("48", 8, 136, 240, 48), ("96", 8, 68, 120, 96)]:
    x = torch.randn(B, H, W, C, device=dev); dy = torch.randn(B, H, W, C, device=dev)
    w = torch.empty(C, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * dy.numel() * C * 9
    line = nm
    for blocks in (256, 384, 512, 768, 1024):
        _lib.lib.catseg_debug_set_wgrad_direct(blocks)
        t = timeit(lambda: ops.conv_bwd_weight(x, dy, w, None, 3, 3, 1, 1, 1))
        line += " | %d: %.1f TF" % (blocks, fl / t / 1e9)
    print(line, flush=True)
