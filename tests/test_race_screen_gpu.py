"""Race screens as tests (tools/stress_dconv3.py and tools/stress_h2.py, shortened): every kernel of the split-precision populations is
deterministic by construction, so the SAME launch repeated -- alone, beside an HBM-heavy neighbour stream and beside an MFMA-heavy one,
into NaN-filled outputs -- must reproduce its first result bit for bit.  This screen found a real LDS-DMA race in round 3 (the first
K-step of dconv3_b3_kernel refilled a weight slot other waves were still reading)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _neighbours(ops, dev):
    big = torch.randn(32, 1024, 1024, device=dev)
    x2 = torch.randn(8, 68, 120, 96, device=dev)
    w2 = (torch.randn(96, 96, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wimg2, y2 = ops.dconv3_weight_image(w2), torch.empty_like(x2)

    def poke(it, side):
        with torch.cuda.stream(side):
            if it % 3 == 1:
                big.mul_(1.0001)                  # HBM-heavy neighbour
            elif it % 3 == 2:                     # another direct kernel beside it (the HRNet branches run concurrently in a step)
                for _ in range(3):
                    ops.dconv3(x2, wimg2, None, out=y2)
    return poke


def _rec(ops, t, slot):
    t._amax = ops.new_amax(t.device)
    t._amax[32 * slot:32 * slot + 1] = t.abs().max().reshape(1).view(torch.int32)
    return t


@pytest.mark.parametrize("shape", [(8, 136, 240, 48), (8, 68, 120, 96), (8, 34, 60, 192), (8, 17, 30, 384), (2, 67, 119, 64), (2, 68, 120, 96)])
def test_direct_trunk_kernels_repeat_bit_for_bit_beside_other_streams(shape):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    B, H, W, C = shape
    dev = torch.device("cuda")
    side = torch.cuda.Stream()
    g = torch.Generator(device="cuda").manual_seed(C)
    x = _rec(ops, torch.randn(B, H, W, C, device=dev, generator=g), 1)
    w = (torch.randn(C, C, 3, 3, device=dev, generator=g) * 0.05).contiguous(memory_format=torch.channels_last)
    poke = _neighbours(ops, dev)
    saved = ops.TRUNK
    ops.TRUNK = "f16x2"
    try:
        imgs = {(h2, dg): ops.dconv3_weight_image(w, backward_data=dg, h2=h2) for h2 in (False, True) for dg in (False, True)}
        dyv = _rec(ops, torch.randn(B, H, W, C, device=dev, generator=g) * 1e-3, 2)

        def run():
            outs = []
            for h2 in (False, True):
                y, (part, nt, _, cnt) = ops.dconv3(x, imgs[(h2, False)], None, out=torch.full_like(x, float("nan")), bn_stats=True,
                                                   x_amax=x._amax if h2 else None)
                outs += [y, part[:3 * nt * C].clone(), cnt.clone()]
                outs.append(ops.dconv3(x, imgs[(h2, True)], out=torch.full_like(x, float("nan")), x_amax=x._amax if h2 else None))
                dw = torch.full_like(w, float("nan"))
                rx, rd = x._amax, dyv._amax
                if not h2:
                    x._amax = dyv._amax = None
                ops.dwgrad3(x, dyv, dw)
                x._amax, dyv._amax = rx, rd
                outs.append(dw)
            return outs
        ref = [t.clone() for t in run()]
        assert all(bool(torch.isfinite(t.float()).all()) for t in ref)
        for it in range(24):
            poke(it, side)
            out = run()
            for i, (a, b) in enumerate(zip(out, ref)):
                assert torch.equal(a, b), ("launch %d, output %d differs from the first run" % (it, i), float((a.float() - b.float()).abs().max()))
        torch.cuda.synchronize()
    finally:
        ops.TRUNK = saved
        ops.release_b3_cache()


@pytest.mark.parametrize("shape", [(8, 136, 240, 720, 512, 3, 1), (2, 33, 47, 208, 264, 3, 1), (1, 17, 19, 224, 256, 1, 0), (3, 9, 11, 208, 520, 3, 1)])
def test_f16x2_head_kernels_repeat_bit_for_bit_beside_other_streams(shape):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    B, H, W, Ci, Co, k, p = shape
    dev = torch.device("cuda")
    side = torch.cuda.Stream()
    saved = (ops.PRECISION, ops.HEADS, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS)
    ops.PRECISION, ops.HEADS = "bf16x3", "f16x2"
    ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS = 1, 16, 16, 1, 1
    try:
        g = torch.Generator(device="cuda").manual_seed(Ci)
        x = torch.randn(B, H, W, Ci, device=dev, generator=g)
        w = (torch.randn(Co, Ci, k, k, device=dev, generator=g) * 0.03).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, H, W, Co, device=dev, generator=g) * 1e-4
        poke = _neighbours(ops, dev)

        def run():
            ops.release_b3_cache()
            ops.PROFILE = []
            y = torch.full((B, H, W, Co), float("nan"), device=dev)
            ops.conv_fwd(x, w, None, Co, k, k, 1, p, 1, out=y, train=True)
            dw = torch.full_like(w, float("nan"))
            ops.conv_bwd_weight(x, dy, dw, None, k, k, 1, p, 1)
            dx = torch.full_like(x, float("nan"))
            ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, 1, p, 1, out=dx)
            kinds = {q[0] for q in ops.PROFILE}
            ops.PROFILE = None
            assert {"fwd_h2", "dgrad_h2", "wgrad_h2"} <= kinds, kinds
            return y, dx, dw
        ref = [t.clone() for t in run()]
        assert all(bool(torch.isfinite(t).all()) for t in ref)
        for it in range(12 if B * H * W > 100000 else 30):
            poke(it, side)
            out = run()
            for i, (a, b) in enumerate(zip(out, ref)):
                assert torch.equal(a, b), ("launch %d, output %d differs from the first run" % (it, i), float((a - b).abs().max()))
        torch.cuda.synchronize()
    finally:
        ops.PROFILE = None
        (ops.PRECISION, ops.HEADS, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS) = saved
        ops.release_b3_cache()
