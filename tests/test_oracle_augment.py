"""oracle/augment.py (numpy restatement of Pillow's GaussianBlur / blend / luma / HSV arithmetic and of torchvision's ColorJitter
operations) against fixtures generated with Pillow itself (tests/golden/make_golden_augment.py): bit-exact."""
import os

import numpy as np

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))


def test_gaussian_blur_matches_pillow():
    from oracle import augment as A
    for r in (3, 4, 5, 6):
        for im, ref in zip(G["imgs"], G["blur_r%d" % r]):
            assert np.array_equal(A.gaussian_blur(im, r), ref)
    for im, ref in zip(G["tiny"], G["tiny_blur_r6"]):
        assert np.array_equal(A.gaussian_blur(im, 6), ref)


def test_color_operations_match_pillow():
    from oracle import augment as A
    for op in range(4):
        i = 0
        while "op%d_f%d" % (op, i) in G.files:
            f = float(G["op%d_f%d" % (op, i)])
            for im, ref in zip(G["imgs"], G["op%d_out%d" % (op, i)]):
                assert np.array_equal(A.adjust(im, op, f), ref), (op, f)
            i += 1
    for im, order, fc, ref in zip(G["imgs"], G["seq_orders"], G["seq_factors"], G["seq_out"]):
        assert np.array_equal(A.color_jitter(im, order, fc), ref)
