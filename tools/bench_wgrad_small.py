"""wgrad tile x split sweep on the small-channel HRNet shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops, _lib
dev = torch.device("cuda")
SHAPES = [("48", 8, 136, 240, 48, 48), ("96", 8, 68, 120, 96, 96), ("192", 8, 34, 60, 192, 192), ("64", 8, 136, 240, 64, 64)]
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, B, H, W, Ci, Co in SHAPES:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, W, Co, device=dev); dw = torch.empty_like(w)
    fl = 2.0 * dy.numel() * Ci * 9
    ref = None
    line = "%-4s" % name
    for (mi, ni) in [(0, 0), (1, 1), (1, 2), (2, 1), (2, 2), (2, 4)]:
        best = (0, 0)
        for sp in ([0] if mi == 0 else [48, 96, 144, 192, 288, 384, 576, 768]):
            _lib.lib.catseg_debug_set_tile(mi, ni); _lib.lib.catseg_debug_set_splits(sp)
            try:
                t = timeit(lambda: ops.conv_bwd_weight(x, dy, dw, None, 3, 3, 1, 1, 1))
            except Exception as e:
                continue
            if ref is None: ref = dw.clone()
            err = float((dw - ref).abs().max() / ref.abs().max())
            tf = fl / t / 1e9
            if tf > best[0]: best = (tf, sp, err)
        line += " | %dx%d %5.1f TF (sp %d, err %.0e)" % (mi, ni, best[0], best[1], best[2] if len(best) > 2 else 0)
    _lib.lib.catseg_debug_set_tile(0, 0); _lib.lib.catseg_debug_set_splits(0)
    print(line, flush=True)
