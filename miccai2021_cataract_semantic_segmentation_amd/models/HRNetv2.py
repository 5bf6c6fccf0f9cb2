"""HRNetv2 on the HIP engine — module tree, state-dict keys and forward of the reference's
models/HRNetv2.py:36-514 (BasicBlock :36-65, Bottleneck :68-106, HighResolutionModule :115-261,
HRNetv2 :264-514).  The reference hard-codes a small configuration (widths 32/64/128/256, one module
per stage); ``config['hrnet']`` may override it (``width``, ``stage1_width``, ``modules``) — the
W48 setting (48/96/192/384, stage-1 width 64, modules 1/4/3) gives the published 65.9 M-parameter net.
BatchNorm momentum 0.01 (:19), bilinear resizes with align_corners=False (:253-256,504-512)."""
import torch
from torch import nn

from .. import engine
from ..engine import BatchNorm2d, Conv2d, EngineNet, add_n, bilinear, image_hw, concat_views, conv_bias, conv_bn_act, tap
from ..utils import num_classes

BN_MOMENTUM = 0.01


def _bn(c):
    return BatchNorm2d(c, momentum=BN_MOMENTUM)


def _conv3x3(cin, cout, stride=1):
    return Conv2d(cin, cout, 3, stride, 1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = _bn(planes)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = _bn(planes)
        self.downsample = downsample
        self.stride = stride

    def run(self, cx, x):
        o = conv_bn_act(cx, x, self.conv1, self.bn1, sole_conv_out=True)      # (o feeds conv2 and nothing else: private_in below)
        idt = x if self.downsample is None else conv_bn_act(cx, x, self.downsample[0], self.downsample[1], relu=False)
        return conv_bn_act(cx, o, self.conv2, self.bn2, relu=True, residual=idt, private_in=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _bn(planes * 4)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample
        self.stride = stride

    def run(self, cx, x):
        o = conv_bn_act(cx, x, self.conv1, self.bn1)
        o = conv_bn_act(cx, o, self.conv2, self.bn2, private_in=True)
        idt = x if self.downsample is None else conv_bn_act(cx, x, self.downsample[0], self.downsample[1], relu=False)
        return conv_bn_act(cx, o, self.conv3, self.bn3, relu=True, residual=idt)


blocks_dict = {"BASIC": BasicBlock, "BOTTLENECK": Bottleneck}


def _run_seq(cx, seq, x):
    """Sequential of blocks, or of (conv, bn[, relu]) groups as the reference builds them"""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (BasicBlock, Bottleneck)):
            x = m.run(cx, x)
            i += 1
        elif isinstance(m, nn.Sequential):
            x = _run_seq(cx, m, x)
            i += 1
        elif isinstance(m, nn.Conv2d):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            x = conv_bn_act(cx, x, m, mods[i + 1], relu=relu)
            i += 3 if relu else 2
        elif isinstance(m, nn.Identity):
            i += 1
        else:
            raise RuntimeError("unexpected module %r" % type(m))
    return x


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError("NUM_BRANCHES does not match NUM_BLOCKS / NUM_CHANNELS / NUM_INCHANNELS")
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = nn.ModuleList([self._make_one_branch(i, blocks, num_blocks, num_channels)
                                       for i in range(num_branches)])
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(inplace=False)

    def _make_one_branch(self, bi, block, num_blocks, num_channels, stride=1):
        downsample = None
        if stride != 1 or self.num_inchannels[bi] != num_channels[bi] * block.expansion:
            downsample = nn.Sequential(Conv2d(self.num_inchannels[bi], num_channels[bi] * block.expansion, 1, stride, bias=False),
                                       _bn(num_channels[bi] * block.expansion))
        layers = [block(self.num_inchannels[bi], num_channels[bi], stride, downsample)]
        self.num_inchannels[bi] = num_channels[bi] * block.expansion
        for _ in range(1, num_blocks[bi]):
            layers.append(block(self.num_inchannels[bi], num_channels[bi]))
        return nn.Sequential(*layers)

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        nb, ch = self.num_branches, self.num_inchannels
        fuse_layers = []
        for i in range(nb if self.multi_scale_output else 1):
            fuse_layer = []
            for j in range(nb):
                if j > i:
                    fuse_layer.append(nn.Sequential(Conv2d(ch[j], ch[i], 1, 1, 0, bias=False), _bn(ch[i])))
                elif j == i:
                    fuse_layer.append(nn.Identity())
                else:
                    convs = []
                    for k in range(i - j):
                        if k == i - j - 1:
                            convs.append(nn.Sequential(Conv2d(ch[j], ch[i], 3, 2, 1, bias=False), _bn(ch[i])))
                        else:
                            convs.append(nn.Sequential(Conv2d(ch[j], ch[j], 3, 2, 1, bias=False), _bn(ch[j]),
                                                       nn.ReLU(inplace=False)))
                    fuse_layer.append(nn.Sequential(*convs))
            fuse_layers.append(nn.ModuleList(fuse_layer))
        return nn.ModuleList(fuse_layers)

    def get_num_inchannels(self):
        return self.num_inchannels

    def run(self, cx, xs):
        if self.num_branches == 1:
            return [_run_seq(cx, self.branches[0], xs[0])]
        # The branches are independent until the fuse layers: one HIP stream each (engine.Ctx.parallel).  The fuse chains
        # (models/HRNetv2.py:237-261: out_i = relu(sum_j f_ij(x_j)); 12 independent chains f_ij) run grouped by their SOURCE branch j ON
        # THAT BRANCH'S STREAM, right behind the branch: chain f_ij needs nothing but x_j, so the region joins only once per module
        # (a join + fork between branches and fuse chains cost one cross-stream hand-over more: 252 -> ~130 idle gaps of ~24 us per
        # step); the gradient of x_j is accumulated by one stream, in a fixed order.  The sums follow on the main stream.
        # (Round 5 measured the alternative placement -- the down-sampling chains f_i0 that start from branch 0 on their DESTINATION branch's
        # stream, with the cross-stream ordering that needs: 118.5 - 120.6 against 111.9 - 112.3 ms eager, 121.7 against 113.5 ms as a graph
        # replay.  The regions are throughput-bound, not bound by branch 0's chain: a kernel trace suggests otherwise -- 75 % of a traced step
        # shows one kernel in flight -- because the tracer serialises dispatches; timings of the untraced step decide.)
        xs = list(xs)
        n_out = len(self.fuse_layers)
        terms = [[None] * self.num_branches for _ in range(n_out)]
        shapes = [tuple(x.shape) for x in xs]       # (branch outputs keep their input's spatial size)
        with cx.parallel(xs[0].device, self.num_branches) as par:
            for j in range(self.num_branches):
                with par.branch(j):
                    xs[j] = _run_seq(cx, self.branches[j], xs[j])
                    for i in range(n_out):
                        if j == i:
                            continue
                        t = _run_seq(cx, self.fuse_layers[i][j], xs[j])
                        terms[i][j] = bilinear(cx, t, shapes[i][1], shapes[i][2], False) if j > i else t
        outs = []
        for i in range(n_out):
            terms[i][i] = xs[i]
            outs.append(add_n(cx, terms[i], relu=True))
        return outs


class HRNetBody(nn.Module):
    """stem + 4 stages; ``run`` returns the four branch outputs of stage 4"""

    def __init__(self, width=32, stage1_width=32, modules=(1, 1, 1)):
        super().__init__()
        w = width
        self.stage1_cfg = {"num_modules": 1, "num_branches": 1, "num_blocks": [4], "num_channels": [stage1_width],
                           "block": "BOTTLENECK", "fuse_method": "SUM"}
        self.stage2_cfg = {"num_modules": modules[0], "num_branches": 2, "num_blocks": [4, 4], "num_channels": [w, 2 * w],
                           "block": "BASIC", "fuse_method": "SUM"}
        self.stage3_cfg = {"num_modules": modules[1], "num_branches": 3, "num_blocks": [4, 4, 4],
                           "num_channels": [w, 2 * w, 4 * w], "block": "BASIC", "fuse_method": "SUM"}
        self.stage4_cfg = {"num_modules": modules[2], "num_branches": 4, "num_blocks": [4, 4, 4, 4],
                           "num_channels": [w, 2 * w, 4 * w, 8 * w], "block": "BASIC", "fuse_method": "SUM"}

    # the members are created by the owning network so that the state-dict keys have no extra prefix


def _make_transition_layer(pre, cur):
    layers = []
    for i in range(len(cur)):
        if i < len(pre):
            if cur[i] != pre[i]:
                layers.append(nn.Sequential(Conv2d(pre[i], cur[i], 3, 1, 1, bias=False), _bn(cur[i]), nn.ReLU(inplace=False)))
            else:
                layers.append(nn.Identity())
        else:
            convs = []
            for j in range(i + 1 - len(pre)):
                cin = pre[-1]
                cout = cur[i] if j == i - len(pre) else cin
                convs.append(nn.Sequential(Conv2d(cin, cout, 3, 2, 1, bias=False), _bn(cout), nn.ReLU(inplace=False)))
            layers.append(nn.Sequential(*convs))
    return nn.ModuleList(layers)


def _make_layer(block, inplanes, planes, blocks, stride=1):
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False), _bn(planes * block.expansion))
    layers = [block(inplanes, planes, stride, downsample)]
    inplanes = planes * block.expansion
    for _ in range(1, blocks):
        layers.append(block(inplanes, planes))
    return nn.Sequential(*layers)


def _make_stage(cfg, num_inchannels, multi_scale_output=True):
    block = blocks_dict[cfg["block"]]
    modules = []
    for i in range(cfg["num_modules"]):
        mso = multi_scale_output or i != cfg["num_modules"] - 1
        modules.append(HighResolutionModule(cfg["num_branches"], block, cfg["num_blocks"], num_inchannels,
                                            cfg["num_channels"], cfg["fuse_method"], mso))
        num_inchannels = modules[-1].get_num_inchannels()
    return nn.Sequential(*modules), num_inchannels


def build_hrnet_trunk(net, width, stage1_width, modules):
    """creates conv1..stage4 as attributes of ``net`` in the reference's construction order; returns the
    channel counts of the four output branches"""
    cfg = HRNetBody(width, stage1_width, modules)
    net.stage1_cfg, net.stage2_cfg, net.stage3_cfg, net.stage4_cfg = cfg.stage1_cfg, cfg.stage2_cfg, cfg.stage3_cfg, cfg.stage4_cfg
    net.conv1 = Conv2d(3, 64, 3, 2, 1, bias=False)
    net.bn1 = _bn(64)
    net.conv2 = Conv2d(64, 64, 3, 2, 1, bias=False)
    net.bn2 = _bn(64)
    net.relu = nn.ReLU(inplace=False)
    block = blocks_dict[net.stage1_cfg["block"]]
    nch = net.stage1_cfg["num_channels"][0]
    net.layer1 = _make_layer(block, 64, nch, net.stage1_cfg["num_blocks"][0])
    pre = [block.expansion * nch]
    for si, name in ((2, "stage2_cfg"), (3, "stage3_cfg"), (4, "stage4_cfg")):
        c = getattr(net, name)
        blk = blocks_dict[c["block"]]
        cur = [n * blk.expansion for n in c["num_channels"]]
        setattr(net, "transition%d" % (si - 1), _make_transition_layer(pre, cur))
        stage, pre = _make_stage(c, cur, True)
        setattr(net, "stage%d" % si, stage)
    # The rounding error of the first layers is what the rest of the trunk amplifies (x 250 from the stem to the logits, x 100 from layer1,
    # x 40 from stage 2, x 12 from stage 3: tools/error_growth.py): the four 3x3 convolutions of layer1 run their forward pass on the fp32
    # kernel with two-level accumulation instead of the two-plane fp16 direct kernel (+0.4 ms per bs-8 step).
    # Round 5: the same holds for the stem's second convolution and the first transition once the gather launches of csrc/pconv1.hip could
    # take their FORWARD pass (tools/error_growth.py at 2 x 3 x 544 x 960: relative RMS error at the stem 2.2e-7 -> 3.7e-7 and 1.4 x at every later tap,
    # 113 instead of 63 label disagreements with fp64 at the logits): they stay on the fp32 kernel forward (backward takes the gather route).
    # CATSEG_EXACT_EARLY = "layer1" / "stem" / "all" (default) selects how far the rule reaches.
    from .. import plan as _plan
    early = _plan.get("exact_early")
    parts = [net.layer1]
    if early in ("stem", "all"):
        parts.append(net.conv2)
    if early == "all":
        parts.append(net.transition1)
    for part in parts:
        for m in part.modules():
            if isinstance(m, Conv2d):
                m.exact_operands = True
    return pre


def run_hrnet_trunk(net, cx, x):
    x = conv_bn_act(cx, x, net.conv1, net.bn1, need_dx=False)
    x = tap("stem", conv_bn_act(cx, x, net.conv2, net.bn2))
    x = tap("layer1", _run_seq(cx, net.layer1, x))
    ys = [x]
    for si in (2, 3, 4):
        trans = getattr(net, "transition%d" % (si - 1))
        nb_prev = len(ys)
        xs = []
        for i, t in enumerate(trans):
            src = ys[i] if i < nb_prev else ys[-1]
            xs.append(src if isinstance(t, nn.Identity) else _run_seq(cx, t, src))
        for mod in getattr(net, "stage%d" % si):
            xs = mod.run(cx, xs)
        ys = xs
        for i, y in enumerate(ys):
            tap("stage%d.b%d" % (si, i), y)
    return ys


def concat_branches(cx, ys, h2_consumers=None):
    """upsample branches 1..3 to branch 0's size (bilinear, align_corners=False) and concatenate.
    h2_consumers: the caller states that the result feeds NOTHING but these Conv2d modules.  When all of them run on blocked f16x2 planes in
    forward and backward-weight (ops.concat_planes_route), a recorded training pass writes the concatenation ONLY as those planes -- the returned
    fp32 tensor is an unwritten placeholder (ops.register_h2_planes) that carries shape and gradient identity."""
    B, H, W, _ = ys[0].shape
    chans = [y.shape[-1] for y in ys]
    cat = torch.empty((B, H, W, sum(chans)), dtype=torch.float32, device=ys[0].device)
    parts, c0 = [], 0
    if h2_consumers and cx.record and cx.train and engine.TAPS is None:
        from .. import ops
        if ops.concat_planes_route(ys, h2_consumers):
            blk, sc = ops.concat_bilinear_h2(ys, H, W)
            ops.register_h2_planes(cat, blk, sc)
            for y, c in zip(ys, chans):
                dst = cat[..., c0:c0 + c]           # (never written: the key under which concat_views hands this slice's gradient out)
                if y.shape[1] == H and y.shape[2] == W:
                    def bwd(src=y, d=dst):
                        g = cx.take(d)
                        if g is not None:
                            cx.give(src, g)
                    cx.push(bwd)
                else:
                    engine.bilinear_backward_of(cx, y, dst, False)
                parts.append((dst, c0, c0 + c))
                c0 += c
            concat_views(cx, cat, parts)
            return cat
    for y, c in zip(ys, chans):
        dst = cat[..., c0:c0 + c]
        if y is ys[0]:
            from .. import ops
            rec = ops.amax_of(y)
            ops.axpy(y, dst, 1.0, False)
            if rec is not None:
                dst._amax = rec         # (a copy: the source's record bounds it; concat_views collects the slices' records)
            part = dst
            if cx.record:
                def bwd(src=y, d=dst):
                    g = cx.take(d)
                    if g is not None:
                        cx.give(src, g)
                cx.push(bwd)
        else:
            part = bilinear(cx, y, H, W, False, out=dst)
        parts.append((part, c0, c0 + c))
        c0 += c
    concat_views(cx, cat, parts)
    return cat


class HRNetv2(EngineNet):
    def __init__(self, config, experiment):
        super().__init__()
        self.num_classes = num_classes(experiment)
        self.pretrained_layers = ["*"]
        self.stem_inplanes = 64
        self.final_conv_kernel = 1
        self.with_head = True
        h = (config or {}).get("hrnet", {}) if isinstance(config, dict) else {}
        pre = build_hrnet_trunk(self, h.get("width", 32), h.get("stage1_width", 32), tuple(h.get("modules", (1, 1, 1))))
        last = int(sum(pre))
        self.last_layer = nn.Sequential(Conv2d(last, last, 1, 1, 0), _bn(last), nn.ReLU(inplace=False),
                                        Conv2d(last, self.num_classes, self.final_conv_kernel, 1,
                                               1 if self.final_conv_kernel == 3 else 0))
        self.out_stride = 4
        self.projector_model = None

    def _body(self, cx, x):
        H, W = image_hw(x)
        ys = run_hrnet_trunk(self, cx, x)
        cat = concat_branches(cx, ys)
        y = conv_bn_act(cx, cat, self.last_layer[0], self.last_layer[1])
        logits = conv_bias(cx, y, self.last_layer[3])
        return [bilinear(cx, logits, H, W, False)]
