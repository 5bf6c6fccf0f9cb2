"""bf16x3 split-precision convolution vs the fp32 MFMA kernel: accuracy against an fp64 CPU reference on small inputs,
speed on the OCRNet-HRNet-W48 / OCRNet-R50 layer shapes (bs 8 @ 544x960)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from miccai2021_cataract_semantic_segmentation_amd import ops, _lib

dev = torch.device("cuda")


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def accuracy(tile=0):
    _lib.lib.catseg_debug_set_b3_tile(tile)
    print("== accuracy vs fp64 (CPU), bf16x3 tile %d ==" % tile)
    for (B, H, W, Ci, Co, k, s, p, d) in [(2, 21, 27, 64, 96, 3, 1, 1, 1), (1, 30, 34, 48, 40, 3, 1, 2, 2), (2, 19, 23, 720, 512, 3, 1, 1, 1),
                                          (2, 24, 24, 256, 256, 1, 1, 0, 1), (2, 33, 29, 96, 192, 3, 2, 1, 1), (1, 16, 20, 24, 25, 1, 1, 0, 1), (2, 40, 44, 16, 512, 3, 1, 1, 1)]:
        g = torch.Generator().manual_seed(Ci + Co)
        x = torch.randn(B, Ci, H, W, generator=g) * torch.exp(2 * torch.randn(B, Ci, 1, 1, generator=g))   # wide dynamic range
        w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5
        b = torch.randn(Co, generator=g)
        y64 = F.conv2d(x.double(), w.double(), b.double(), s, p, d)
        gy = torch.randn(y64.shape, generator=g)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        wd = w.to(dev).contiguous(memory_format=torch.channels_last)
        y32 = ops.conv_fwd(xd, wd, b.to(dev), Co, k, k, s, p, d)
        yb3 = ops.conv_fwd_b3(tuple(xd.shape), ops.split3(xd), ops.split3_weight(wd), b.to(dev), Co, k, k, s, p, d)
        ref = y64.permute(0, 2, 3, 1)
        sc = float(ref.abs().max())
        e32 = float((y32.cpu().double() - ref).abs().max()) / sc
        eb3 = float((yb3.cpu().double() - ref).abs().max()) / sc
        line = "fwd %s: max err / scale  fp32 kernel %.3g   bf16x3 %.3g" % ((B, H, W, Ci, Co, k, s, p, d), e32, eb3)
        if s == 1:
            xr = x.double().requires_grad_()
            F.conv2d(xr, w.double(), None, s, p, d).backward(gy.double())
            gyd = ops.new_act(B, y64.shape[2], y64.shape[3], Co, dev, zero=True)
            gyd.copy_(gy.permute(0, 2, 3, 1))
            d32 = ops.conv_bwd_data(gyd, wd, tuple(xd.shape), k, k, s, p, d)
            db3 = ops.conv_bwd_data_b3(ops.split3(gyd), ops.split3_weight_t(wd), tuple(xd.shape), Co, k, k, s, p, d)
            rd = xr.grad.permute(0, 2, 3, 1)
            scd = float(rd.abs().max())
            line += " | dgrad fp32 %.3g bf16x3 %.3g" % (float((d32.cpu().double() - rd).abs().max()) / scd, float((db3.cpu().double() - rd).abs().max()) / scd)
        print(line, flush=True)


SHAPES = [("head 3x3 720>512 @136x240", 8, 136, 240, 720, 512, 3, 1, 1, 1), ("ocr 1x1 1024>512", 8, 136, 240, 1024, 512, 1, 1, 0, 1),
          ("ocr 1x1 512>256", 8, 136, 240, 512, 256, 1, 1, 0, 1), ("branch 3x3 384>384 @17x30", 8, 17, 30, 384, 384, 3, 1, 1, 1),
          ("branch 3x3 192>192 @34x60", 8, 34, 60, 192, 192, 3, 1, 1, 1), ("branch 3x3 96>96 @68x120", 8, 68, 120, 96, 96, 3, 1, 1, 1),
          ("branch 3x3 48>48 @136x240", 8, 136, 240, 48, 48, 3, 1, 1, 1), ("r50 l4 3x3d4 512>512", 8, 68, 120, 512, 512, 3, 1, 4, 4),
          ("r50 high_map 3x3 2048>512", 8, 68, 120, 2048, 512, 3, 1, 1, 1), ("r50 l4 1x1 2048>512", 8, 68, 120, 2048, 512, 1, 1, 0, 1),
          ("r50 l3 1x1 256>1024", 8, 68, 120, 256, 1024, 1, 1, 0, 1)]


def _wgrad32(x, dy, dw, k, s, p, d):
    import ctypes
    desc = ops.make_desc(x.shape, ops.ld_of(x), dy.shape[-1], ops.ld_of(dy), k, k, s, p, d)
    ws = ops.workspace(_lib.lib.catseg_conv2d_bwd_weight_workspace(ctypes.byref(desc)), x.device)
    _lib.check(_lib.lib.catseg_conv2d_bwd_weight(ctypes.byref(desc), ops.ptr(x), ops.ptr(dy), ops.ptr(dw), 0, ops.ptr(ws), ws.numel(), ops.stream()))


def _wgradb3(x, dy, dw, xp, dyp, Co, Ci, k, s, p, d):
    import ctypes
    desc = ops.make_desc(x.shape, Ci, Co, (Co + 7) // 8 * 8, k, k, s, p, d)
    ws = ops.workspace(_lib.lib.catseg_conv2d_bwd_weight_bf16x3_workspace(ctypes.byref(desc)) + 1024, x.device)
    _lib.check(_lib.lib.catseg_conv2d_bwd_weight_bf16x3(ctypes.byref(desc), ops.ptr(xp), ops.ptr(dyp), ops.ptr(dw), ops.ptr(ws), ws.numel(), ops.stream()))


def speed(tiles):
    print("== speed (TFLOP/s-equivalent = 2MNK / time) ==")
    for name, B, H, W, Ci, Co, k, s, p, d in SHAPES:
        x = torch.randn(B, H, W, Ci, device=dev)
        w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        y = ops.conv_fwd(x, w, None, Co, k, k, s, p, d)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        fl = 2.0 * y.numel() * Ci * k * k
        t32 = timeit(lambda: ops.conv_fwd(x, w, None, Co, k, k, s, p, d, out=y))
        td32 = timeit(lambda: ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, s, p, d, out=dx))
        xp, wp = ops.split3(x), ops.split3_weight(w)
        dyp, wtp = ops.split3(dy), ops.split3_weight_t(w)
        dw = torch.empty_like(w)
        tw32 = timeit(lambda: ops.conv_bwd_weight.__wrapped__(x, dy, dw, None, k, k, s, p, d) if hasattr(ops.conv_bwd_weight, "__wrapped__") else _wgrad32(x, dy, dw, k, s, p, d))
        twb = timeit(lambda: _wgradb3(x, dy, dw, xp, dyp, Co, Ci, k, s, p, d))
        tsx = timeit(lambda: ops.split3(x))
        tsw = timeit(lambda: ops.split3_weight(w))
        res = []
        for t in tiles:
            _lib.lib.catseg_debug_set_b3_tile(t)
            tb = timeit(lambda: ops.conv_fwd_b3(tuple(x.shape), xp, wp, None, Co, k, k, s, p, d, out=y))
            tdb = timeit(lambda: ops.conv_bwd_data_b3(dyp, wtp, tuple(x.shape), Co, k, k, s, p, d, out=dx))
            res.append("t%d: fwd %.3f ms %.0f TF, dgrad %.3f ms %.0f TF" % (t, tb, fl / tb / 1e9, tdb, fl / tdb / 1e9))
        _lib.lib.catseg_debug_set_b3_tile(0)
        print("%-28s %6.1f GF | fp32 fwd %.3f ms %.0f TF dgrad %.3f ms %.0f TF wgrad %.3f ms %.0f TF | b3 wgrad %.3f ms %.0f TF | split x %.3f w %.3f ms | %s"
              % (name, fl / 1e9, t32, fl / t32 / 1e9, td32, fl / td32 / 1e9, tw32, fl / tw32 / 1e9, twb, fl / twb / 1e9, tsx, tsw, " ; ".join(res)), flush=True)


if __name__ == "__main__":
    accuracy()
    accuracy(9)
    _lib.lib.catseg_debug_set_b3_tile(0)
    speed([int(t) for t in sys.argv[1:]] or [0])
