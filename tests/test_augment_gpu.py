"""csrc/augment.hip (GaussianBlur, the four ColorJitter operations, uint8 flip + reflect pad) through the C ABI: bit-exact against the
Pillow-generated fixtures (tests/golden/augment.npz) and against oracle/augment.py on larger inputs, incl. every RGB triple."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_gaussian_blur_matches_pillow_fixtures():
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.utils import GpuAugment
    aug = GpuAugment()
    imgs = torch.from_numpy(G["imgs"]).cuda()
    for r in (3, 4, 5, 6):
        out = aug.blur(imgs, [r, r, r])
        assert np.array_equal(out.cpu().numpy(), G["blur_r%d" % r])
    mixed = aug.blur(imgs, [0, 5, 3])                                   # per-image radii, 0 = untouched
    assert np.array_equal(mixed[0].cpu().numpy(), G["imgs"][0])
    assert np.array_equal(mixed[1].cpu().numpy(), G["blur_r5"][1]) and np.array_equal(mixed[2].cpu().numpy(), G["blur_r3"][2])
    assert np.array_equal(imgs.cpu().numpy(), G["imgs"])                # the input is not modified
    tiny = aug.blur(torch.from_numpy(G["tiny"]).cuda(), [6, 6])         # lines shorter than the window
    assert np.array_equal(tiny.cpu().numpy(), G["tiny_blur_r6"])


def test_color_operations_match_pillow_fixtures():
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.utils import GpuAugment
    aug = GpuAugment()
    imgs = torch.from_numpy(G["imgs"]).cuda()
    B = imgs.shape[0]
    for op in range(4):
        i = 0
        while "op%d_f%d" % (op, i) in G.files:
            f = float(G["op%d_f%d" % (op, i)])
            factors = np.zeros((B, 4))
            factors[:, op] = f
            out = aug.color_jitter(imgs, np.full((B, 1), op), factors)
            assert np.array_equal(out.cpu().numpy(), G["op%d_out%d" % (op, i)]), (op, f)
            i += 1
    out = aug.color_jitter(imgs, G["seq_orders"], G["seq_factors"])      # a different permutation and factors per image
    assert np.array_equal(out.cpu().numpy(), G["seq_out"])


def test_hue_and_saturation_on_every_rgb_triple_vs_oracle():
    _need_gpu()
    from oracle import augment as A
    from miccai2021_cataract_semantic_segmentation_amd.utils import GpuAugment
    aug = GpuAugment()
    r, g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    rgb = np.stack([r.ravel(), g.ravel(), b.ravel()], 1).reshape(1, 4096, 4096, 3)
    dev = torch.from_numpy(rgb).cuda()
    for op, f in ((A.HUE, 0.05), (A.HUE, -0.037), (A.SATURATION, 1.5), (A.SATURATION, 2 / 3), (A.BRIGHTNESS, 1.37), (A.CONTRAST, 0.71)):
        factors = np.zeros((1, 4))
        factors[0, op] = f
        out = aug.color_jitter(dev, [[op]], factors).cpu().numpy()[0]
        assert np.array_equal(out, A.adjust(rgb[0], op, f)), (op, f)


def test_blur_and_jitter_batch_vs_oracle_and_ingest_order():
    _need_gpu()
    from oracle import augment as A
    from miccai2021_cataract_semantic_segmentation_amd.utils import GpuIngest, sample_blur, sample_color_jitter
    rng = np.random.default_rng(3)
    B, H, W = 4, 60, 84
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    lbl = rng.integers(0, 36, (B, H, W), dtype=np.uint8)
    flips = np.array([0, 1, 2, 3], dtype=np.int32)
    radii = np.array([0, 3, 6, 4], dtype=np.int32)
    orders, factors = sample_color_jitter(B, generator=torch.Generator().manual_seed(5))
    assert sorted(orders[0].tolist()) == [0, 1, 2, 3] and (factors[:, :3] >= 2 / 3).all() and (np.abs(factors[:, 3]) <= 0.05).all()
    r2 = sample_blur(1000, random=np.random.RandomState(0))
    assert set(np.unique(r2)) <= {0, 3, 4, 5, 6} and 20 <= (r2 > 0).sum() <= 90
    ing = GpuIngest(3, pad=(2, 2))
    x, labels = ing(torch.from_numpy(img), torch.from_numpy(lbl), flips, blur_radii=radii, jitter=(orders, factors))
    x_plain, labels_plain = ing(torch.from_numpy(img), torch.from_numpy(lbl), flips)
    assert torch.equal(labels, labels_plain)                            # labels are not touched by the image augmentations
    for b in range(B):
        ref = img[b]
        if flips[b] & 2:
            ref = ref[::-1]
        if flips[b] & 1:
            ref = ref[:, ::-1]
        ref = np.pad(ref, ((2, 2), (0, 0), (0, 0)), mode="reflect")      # PadNP(ver=(2, 2), hor=(0, 0), 'reflect')
        if radii[b] > 0:
            ref = A.gaussian_blur(ref, int(radii[b]))
        ref = A.color_jitter(ref, orders[b], factors[b])
        want = torch.from_numpy(np.ascontiguousarray(ref)).permute(2, 0, 1).float() / 255.0   # ToTensor
        assert torch.equal(x[b].cpu(), want), b
