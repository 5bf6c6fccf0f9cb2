"""direct backward-weight: launch time against the number of blocks (= partial-sum slabs x variants), catseg_debug_set_dwgrad3_blocks"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd._lib import lib
dev = torch.device("cuda")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, C) in [(8, 136, 240, 48), (8, 68, 120, 96), (8, 34, 60, 192), (8, 17, 30, 384), (8, 136, 240, 64)]:
    x = torch.randn(B, H, W, C, device=dev); dy = torch.randn(B, H, W, C, device=dev)
    dw = torch.empty(C, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    out = "C=%3d" % C
    for rnd in range(2):
        for nb in (256, 384, 512, 768, 1024):
            lib.catseg_debug_set_dwgrad3_blocks(nb)
            out += "  %d: %5.1f" % (nb, timeit(lambda: ops.dwgrad3(x, dy, dw)))
        out += " |"
    print(out, flush=True)
lib.catseg_debug_set_dwgrad3_blocks(0)
