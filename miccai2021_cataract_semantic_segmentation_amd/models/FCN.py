"""FCN-8s on the HIP engine -- constructor / forward contract and state-dict keys of the reference's models/FCN.py:7-61 (Long et al.:
seven conv + ReLU layers with 2x2 max-pooling, 1x1 class scores from pool3 / pool4 / conv7 fused through learnt ConvTranspose2d
up-sampling by 2, 2 and 8).  The input height and width must be multiples of 32 (the reference's additions at :57,60 need equal shapes)."""
import numpy as np

from ..engine import Conv2d, ConvTranspose2d, EngineNet, add_classes, conv_act, conv_bias, conv_transpose, image_hw, maxpool2
from ..utils import num_classes


def _padded_conv2d(cin, cout, k):
    # utils/torch_utils.py:130-148 of the reference (stride 1, dilation 1): padding = k // 2
    return Conv2d(int(cin), int(cout), k, 1, (k - 1 + 1) // 2)


def _padded_convtranspose2d(cin, cout, k, stride):
    # utils/torch_utils.py:151-168 of the reference (output_padding 0): padding = (k - stride + 1) // 2  ->  output = stride * input
    return ConvTranspose2d(int(cin), int(cout), k, stride, (k - stride + 1) // 2)


class FCN(EngineNet):
    def __init__(self, config, experiment):
        super().__init__()
        self.num_classes = num_classes(experiment)
        self.width = config["width"]
        n_ch = np.round(np.array([64, 128, 256, 512, 512, 1024, 1024]) * self.width).astype("i")
        if any(int(c) % 4 for c in n_ch):
            raise NotImplementedError("FCN width %r: channel counts %s must be multiples of 4 on the HIP engine" % (self.width, list(n_ch)))
        K = self.num_classes
        self.conv1 = _padded_conv2d(3, n_ch[0], 3)
        self.conv2 = _padded_conv2d(n_ch[0], n_ch[1], 3)
        self.conv3 = _padded_conv2d(n_ch[1], n_ch[2], 3)
        self.conv4 = _padded_conv2d(n_ch[2], n_ch[3], 3)
        self.conv5 = _padded_conv2d(n_ch[3], n_ch[4], 3)
        self.conv6 = _padded_conv2d(n_ch[4], n_ch[5], 3)
        self.conv7 = _padded_conv2d(n_ch[5], n_ch[6], 1)
        self.conv8 = _padded_conv2d(n_ch[6], K, 1)
        self.p4_conv = _padded_conv2d(n_ch[3], K, 1)
        self.deconv32 = _padded_convtranspose2d(K, K, 4, 2)
        self.p3_conv = _padded_conv2d(n_ch[2], K, 1)
        self.deconv16 = _padded_convtranspose2d(K, K, 4, 2)
        self.deconv8 = _padded_convtranspose2d(K, K, 16, 8)

    def _body(self, cx, x):
        H, W = image_hw(x)
        if H % 32 or W % 32:
            raise ValueError("FCN: input %d x %d is not a multiple of 32 (models/FCN.py:57 of the reference adds a 2x up-sampled "
                             "pool5 score to the pool4 score)" % (H, W))
        p1 = maxpool2(cx, conv_act(cx, x, self.conv1))
        p2 = maxpool2(cx, conv_act(cx, p1, self.conv2))
        p3 = maxpool2(cx, conv_act(cx, p2, self.conv3))
        p4 = maxpool2(cx, conv_act(cx, p3, self.conv4))
        p5 = maxpool2(cx, conv_act(cx, p4, self.conv5))
        c7 = conv_act(cx, conv_act(cx, p5, self.conv6), self.conv7)
        dc32 = conv_transpose(cx, conv_bias(cx, c7, self.conv8), self.deconv32)
        fcn_16s = add_classes(cx, dc32, conv_bias(cx, p4, self.p4_conv))
        dc16 = conv_transpose(cx, fcn_16s, self.deconv16)
        fcn_8s = add_classes(cx, dc16, conv_bias(cx, p3, self.p3_conv))
        return [conv_transpose(cx, fcn_8s, self.deconv8)]
