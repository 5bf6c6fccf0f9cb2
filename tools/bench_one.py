"""Run one conv shape repeatedly (for rocprofv3 --pmc passes).
usage: bench_one.py [fwd|dgrad|wgrad|b3fwd|b3dgrad|b3fwdblk|b3dgradblk|b3wgrad|h2fwd|h2dgrad|h2wgrad] [n] [B,H,W,Ci,Co,k,s,p,d] [b3 tile]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
B, H, W, Ci, Co, k, s, p, d = 8, 68, 120, 512, 512, 3, 1, 4, 4
if len(sys.argv) > 3:
    B, H, W, Ci, Co, k, s, p, d = [int(v) for v in sys.argv[3].split(",")]
x = torch.randn(B, H, W, Ci, device=dev)
w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
y = ops.conv_fwd(x, w, None, Co, k, k, s, p, d)
dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
if kind.startswith("b3"):
    from miccai2021_cataract_semantic_segmentation_amd import _lib
    if len(sys.argv) > 4:
        _lib.lib.catseg_debug_set_b3_tile(int(sys.argv[4]))
    xp, wp, dyp, wtp = ops.split3(x), ops.split3_weight(w), ops.split3(dy), ops.split3_weight_t(w)
    if kind.endswith("blk"):
        xb, wb, dyb, wtb = ops.split3_blocked(x)[0], ops.split3_weight_blocked(w), ops.split3_blocked(dy)[0], ops.split3_weight_t_blocked(w)
if kind.startswith("h2"):
    import ctypes
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    xb, xpl, xsc = ops.split2h(x, True, True)
    gb, gpl, gsc = ops.split2h(dy, True, True)
    wb, wsc = ops.split2h_weight_blocked(w)
    wtb, wtsc = ops.split2h_weight_t_blocked(w)
    dfw = ops.make_desc(x.shape, Ci, Co, ops.ld_of(y), k, k, s, p, d)
    dbw = ops.make_desc(x.shape, Ci, Co, (Co + 7) // 8 * 8, k, k, s, p, d)
    ws = torch.empty(max(lib.catseg_conv2d_bwd_weight_f16x2_workspace(ctypes.byref(dbw)), 256), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for _ in range(n):
    if kind == "h2fwd":
        ops.check(lib.catseg_conv2d_fwd_f16x2_blocked(ctypes.byref(dfw), ops.ptr(xb), ops.ptr(xsc), ops.ptr(wb), ops.ptr(wsc), None, ops.ptr(y), 0,
                                                      None, 0, None, None, ops.stream()))
    elif kind == "h2dgrad":
        ops.check(lib.catseg_conv2d_bwd_data_f16x2_blocked(ctypes.byref(dbw), ops.ptr(gb), ops.ptr(gsc), ops.ptr(wtb), ops.ptr(wtsc), ops.ptr(dx), 0,
                                                           ops.stream()))
    elif kind == "h2wgrad":
        ops.check(lib.catseg_conv2d_bwd_weight_f16x2(ctypes.byref(dbw), ops.ptr(xpl), ops.ptr(xsc), ops.ptr(gpl), ops.ptr(gsc), ops.ptr(dw),
                                                     ops.ptr(ws), ws.numel(), ops.stream()))
    elif kind == "b3fwd": ops.conv_fwd_b3(tuple(x.shape), xp, wp, None, Co, k, k, s, p, d, out=y)
    elif kind == "b3fwdblk": ops.conv_fwd_b3_blocked(tuple(x.shape), xb, wb, None, Co, k, k, s, p, d, out=y)
    elif kind == "b3dgradblk": ops.conv_bwd_data_b3_blocked(dyb, wtb, tuple(x.shape), Co, k, k, s, p, d, out=dx)
    elif kind == "b3dgrad": ops.conv_bwd_data_b3(dyp, wtp, tuple(x.shape), Co, k, k, s, p, d, out=dx)
    elif kind == "fwd": ops.conv_fwd(x, w, None, Co, k, k, s, p, d, out=y)
    elif kind == "dgrad": ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, s, p, d, out=dx)
    else: ops.conv_bwd_weight(x, dy, dw, None, k, k, s, p, d)
torch.cuda.synchronize()
print("done", kind, 2.0 * y.numel() * Ci * k * k / 1e9, "GF per launch")
