"""16x16x4 'narrow' igemm forms vs the 32x32x2 tiles on the HRNet branch shapes (48 / 96 / 192 channels): correctness vs the
default path and TFLOP/s.  Many launches per timing so that launch overhead does not dominate."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops, _lib
dev = torch.device("cuda")
SHAPES = [("48>48", 8, 136, 240, 48, 48), ("96>96", 8, 68, 120, 96, 96), ("192>192", 8, 34, 60, 192, 192), ("48>96s1", 8, 136, 240, 48, 96)]
FWD = [(0, 0), (1, 1), (2, 1), (2, 2), (16 + 2, 1), (16 + 4, 1), (16 + 2, 2), (16 + 4, 2)]
WG = [(0, 0), (1, 1), (1, 2), (32 + 1, 2), (32 + 1, 4), (32 + 2, 2), (32 + 2, 4)]
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def name(mi, ni):
    return "%s%dx%d" % ({0: "", 1: "nN", 2: "nM"}[mi >> 4], mi & 15, ni)
for nm, B, H, W, Ci, Co in SHAPES:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, Co, 3, 3, 1, 1, 1)
    dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * y.numel() * Ci * 9
    refs = {}
    for kind, tiles in (("fwd", FWD), ("dgrad", FWD), ("wgrad", WG)):
        line = "%-8s %-5s" % (nm, kind)
        for (mi, ni) in tiles:
            _lib.lib.catseg_debug_set_tile(mi, ni)
            try:
                if kind == "fwd":
                    f = lambda: ops.conv_fwd(x, w, None, Co, 3, 3, 1, 1, 1, out=y); out = y
                elif kind == "dgrad":
                    f = lambda: ops.conv_bwd_data(dy, w, tuple(x.shape), 3, 3, 1, 1, 1, out=dx); out = dx
                else:
                    f = lambda: ops.conv_bwd_weight(x, dy, dw, None, 3, 3, 1, 1, 1); out = dw
                t = timeit(f)
            except Exception as e:
                line += " | %s ERR" % name(mi, ni); continue
            if kind not in refs: refs[kind] = out.clone()
            err = float((out - refs[kind]).abs().max() / refs[kind].abs().max())
            line += " | %s %5.1f%s" % (name(mi, ni), fl / t / 1e9, "" if err < 1e-5 else " BAD(%.0e)" % err)
        _lib.lib.catseg_debug_set_tile(0, 0)
        print(line, flush=True)
