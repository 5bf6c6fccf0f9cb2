"""Execution engine: explicit tape of native kernels (no tracing compiler, no torch ops in the
hot path).  A forward pass records one backward closure per fused layer; the backward pass
walks the tape in reverse, writing parameter gradients straight into one flat fp32 buffer
(so the optimiser is a single kernel and data-parallel all-reduce works on contiguous buckets).

torch.autograd sees exactly one node per network (``NetFunction``) and one per loss, which keeps
the reference's ``loss.backward(); optimiser.step()`` calling convention working.
"""
import torch
from torch import nn

from . import ops
from . import plan as _plan


# --------------------------------------------------------------------------- parameter containers
class Conv2d(nn.Conv2d):
    """Parameter container with nn.Conv2d's init / state-dict behaviour; runs on the HIP engine."""
    stem = False
    exact_operands = False      # forward on the fp32 MFMA kernel whatever the split-precision kernels could take: ops.conv_fwd(exact=)

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("engine Conv2d is executed by the owning network, not called directly")


class BatchNorm2d(nn.BatchNorm2d):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._pending_batches = 0  # num_batches_tracked is flushed lazily (no per-step kernel)

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("engine BatchNorm2d is executed by the owning network, not called directly")


def flush_bn_counters(module):
    for m in module.modules():
        if isinstance(m, BatchNorm2d) and m._pending_batches:
            m.num_batches_tracked += m._pending_batches
            m._pending_batches = 0


# --------------------------------------------------------------------------- flat parameter storage
class FlatParams:
    """All parameters of a network as views into one flat buffer (conv weights physically OHWI),
    gradients likewise.  Rebuilt automatically if the module was moved (.to / .cuda)."""
    ALIGN = 64  # floats

    def __init__(self, module):
        self.module = module
        self.params = [p for p in module.parameters()]
        self.device = None
        self.flat = self.grad = None
        self.offsets = {}

    def _stale(self):
        if self.flat is None:
            return True
        p0, pn = self.params[0], self.params[-1]
        return (p0.device != self.flat.device or p0.data_ptr() != self.flat.data_ptr() + 4 * self.offsets[id(p0)]
                or pn.data_ptr() != self.flat.data_ptr() + 4 * self.offsets[id(pn)])

    def ensure(self):
        if not self._stale():
            return self
        dev = self.params[0].device
        off = 0
        offs = {}
        for p in self.params:
            offs[id(p)] = off
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p in self.params:
            o, n = offs[id(p)], p.numel()
            pv, gv = self._view(flat, o, p), self._view(grad, o, p)
            pv.copy_(p.data)
            p.data = pv
            p.grad = gv
        self.flat, self.grad, self.offsets, self.device, self.numel = flat, grad, offs, dev, off
        return self

    @staticmethod
    def _view(buf, o, p):
        if p.dim() == 4:
            O, I, kh, kw = p.shape
            return buf[o:o + p.numel()].view(O, kh, kw, I).permute(0, 3, 1, 2)
        return buf[o:o + p.numel()].view(p.shape)

    def bind_grads(self):
        """(re)attach .grad views (optimizer.zero_grad(set_to_none=True) detaches them)"""
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * self.offsets[id(p)]:
                p.grad = self._view(self.grad, self.offsets[id(p)], p)


# --------------------------------------------------------------------------- tape
_side_streams = {}


def side_streams(device, n):
    """a small pool of HIP streams per device for the parallel-branch regions (HRNet's branches are independent)"""
    key = (device.type, device.index)
    pool = _side_streams.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


PREP_ASYNC = _plan.get("prep_async")
_prep_streams = {}


def prep_stream(device):
    key = (device.type, device.index)
    if key not in _prep_streams:
        _prep_streams[key] = torch.cuda.Stream(device=device)
    return _prep_streams[key]


PARALLEL_BRANCHES = True   # run independent branches (HRNet stages) on separate HIP streams, forward and backward
# streams a parallel region spreads its branches over (branch i runs on stream i % BRANCH_STREAMS): A/B knob, CATSEG_BRANCH_STREAMS
BRANCH_STREAMS = _plan.get("branch_streams")
LAST_BRANCH_ON_MAIN = _plan.get("last_branch_on_main")


class _Region:
    """tape marker of a parallel region: the streams its branches ran on"""
    def __init__(self, streams):
        self.streams = streams


class Ctx:
    def __init__(self, train, record, on_param_grad=None):
        self.train = train
        self.record = record
        self.tape = []
        self.grads = {}
        self.shared = set()
        self.on_param_grad = on_param_grad
        self.on_quiet = None           # backward: called on the launch stream between two tape entries outside every parallel region
        self.claimed = set()
        self.branch_stream = None      # stream of the branch being recorded (None: the main stream)
        self.region = None
        self._region_depth = 0         # backward: parallel regions entered and not yet joined
        self._deferred = []            # backward: parameters whose 'gradient ready' signal waits for the join
        self.bn_src = {}               # id(z) -> (y, stats, gamma, beta) of a conv_bn_act output z = relu(bn(y))
        self.bn_pre = {}               # backward: id(z) -> per-tile sums of the already masked gradient of z (conv_bn_act private_in)

    def claim(self, *params):
        """every parameter may feed exactly one recorded layer: the backward tape WRITES (does not accumulate) parameter
        gradients and the data-parallel reducer launches a bucket after one 'ready' signal per parameter, so a module
        applied twice in one forward would silently lose a gradient term"""
        if not self.record:
            return
        for p in params:
            if p is None:
                continue
            if id(p) in self.claimed:
                raise NotImplementedError("a parameterised layer is applied twice in one forward pass (shared module): "
                                          "not supported by the HIP engine's backward tape")
            self.claimed.add(id(p))

    def push(self, fn):
        if self.record:
            self.tape.append((fn, self.branch_stream))

    # ---- parallel regions: independent branches on separate HIP streams --------------------------------------------------
    # The kernels of a 192- or 384-channel HRNet branch fill a fraction of the 256 CUs; its three sibling branches are
    # independent until the fuse layer.  Forward: every branch runs on its own side stream (all of them wait for the main
    # stream at the region's start, the main stream waits for all of them at its end and launches nothing in between, so that
    # the caching allocator never hands a block that a side stream still reads to main-stream work).  The tape remembers each
    # closure's stream and the region's boundaries; the backward replay mirrors the same fork / join.
    def parallel(self, device, n):
        cx = self

        class _Par:
            def __enter__(self_):
                self_.on = PARALLEL_BRANCHES and n > 1 and device.type == "cuda"
                if not self_.on:
                    return self_
                self_.main = torch.cuda.current_stream(device)
                ops.images_ready()              # (the branch streams fork from here: they must not wait on the prep stream themselves)
                ns = max(1, min(n, BRANCH_STREAMS))
                if LAST_BRANCH_ON_MAIN and ns >= 4 and cx.record:
                    # The runtime spreads streams over FOUR hardware queues: the main stream holds one, so of four side streams two share a
                    # queue and run one after the other (rocprofv3 kernel trace: streams 3 and 4 on queue 4; per-branch stream times of a
                    # stage-4 module 4.4 / 4.1 / 5.4 / 5.4 ms backward).  The main stream idles during a region: the last branch runs on it.
                    # Only for a RECORDED pass: its tape keeps every tensor a side stream reads alive until the backward has used it, so that
                    # main-stream allocations inside the region cannot be handed a block a side stream still reads (an inference pass frees
                    # a module's inputs as it goes: there the main stream launches nothing between fork and join, as before).
                    self_.streams = side_streams(device, ns - 1) + [self_.main]
                else:
                    self_.streams = side_streams(device, ns)
                ev = torch.cuda.Event(enable_timing=MARKS is not None)
                ev.record(self_.main)
                self_.t0 = ev
                for st in self_.streams:
                    st.wait_event(ev)
                cx.region = _Region(self_.streams)
                if cx.record:
                    cx.tape.append(("region_begin", cx.region))
                return self_

            def branch(self_, i):
                return _Branch(self_, i)

            def __exit__(self_, *exc):
                if not self_.on:
                    return False
                ends = []
                for st in self_.streams:
                    ev = torch.cuda.Event(enable_timing=MARKS is not None)
                    ev.record(st)
                    self_.main.wait_event(ev)
                    ends.append(ev)
                if MARKS is not None:
                    MARKS.append(("region_fwd", self_.t0, ends))
                if cx.record:
                    cx.tape.append(("region_end", cx.region))
                cx.region = None
                return False

        class _Branch:
            def __init__(self_, par, i):
                self_.par, self_.i = par, i

            def __enter__(self_):
                if self_.par.on:
                    st = self_.par.streams[self_.i % len(self_.par.streams)]
                    self_.ctxm = torch.cuda.stream(st)
                    self_.ctxm.__enter__()
                    cx.branch_stream = st
                return self_

            def __exit__(self_, *exc):
                if self_.par.on:
                    cx.branch_stream = None
                    self_.ctxm.__exit__(*exc)
                return False

        return _Par()

    def take(self, t):
        self.shared.discard(id(t))
        return self.grads.pop(id(t), None)

    def _own(self, t):
        """a gradient buffer that several activations share (fan-out of an add) is copied before
        anything is accumulated into it"""
        k = id(t)
        if k in self.shared:
            self.shared.discard(k)
            src = self.grads[k]
            own = torch.empty(src.shape, dtype=torch.float32, device=src.device)
            ops.axpy(src, own, 1.0, False)
            self.grads[k] = own
        return ops.drop_amax(self.grads[k])     # (the caller accumulates into it: a producer's amax record no longer bounds it)

    def give(self, t, g, shared=False):
        assert id(t) not in self.bn_pre, "conv_bn_act(private_in=True): the input has a second consumer"
        if id(t) not in self.grads:
            self.grads[id(t)] = g
            if shared:
                self.shared.add(id(t))
        else:
            ops.axpy(g, self._own(t), 1.0, True)

    def dest(self, t):
        """buffer the gradient of activation t must be written to: (buffer, accumulate?)"""
        assert id(t) not in self.bn_pre, "conv_bn_act(private_in=True): the input has a second consumer"
        if id(t) in self.grads:
            return self._own(t), True
        C = t.shape[-1]
        ld = ops.ld_of(t) if t.dim() == 4 else C
        if t.dim() == 4 and ld != C and ld <= 64:   # small padded rows (class logits): keep the zero pad
            buf = ops.new_act(t.shape[0], t.shape[1], t.shape[2], C, t.device, ld=ld, zero=True)
        else:
            buf = torch.empty(t.shape, dtype=torch.float32, device=t.device)
        self.grads[id(t)] = buf
        return buf, False

    def pgrad(self, p):
        return p.grad

    def done(self, *params):
        """gradient of these parameters has been enqueued.  Inside a parallel region the signal is held back until the region
        has joined: the data-parallel reducer launches a bucket's all-reduce behind the CURRENT stream only, and the other
        members of the bucket may have been written on sibling streams."""
        if self.on_param_grad is not None:
            for p in params:
                if p is not None:
                    if self._region_depth > 0:
                        self._deferred.append(p)
                    else:
                        self.on_param_grad(p)

    def backward(self):
        tape = self.tape
        main = None
        region_t0 = None
        while tape:
            fn, tag = tape.pop()
            if isinstance(fn, str):
                region = tag
                if main is None:
                    main = torch.cuda.current_stream(region.streams[0].device)
                if fn == "region_end":          # (reverse order) entering the region: the side streams wait for the main stream
                    ev = torch.cuda.Event(enable_timing=MARKS is not None)
                    ev.record(main)
                    for st in region.streams:
                        st.wait_event(ev)
                    self._region_depth += 1
                    region_t0 = ev
                else:                           # leaving it: the main stream waits for every branch
                    ends = []
                    for st in region.streams:
                        ev = torch.cuda.Event(enable_timing=MARKS is not None)
                        ev.record(st)
                        main.wait_event(ev)
                        ends.append(ev)
                    if MARKS is not None:       # (tools/stage_times.py: how long each branch stream of this region's backward ran)
                        MARKS.append(("region_bwd", region_t0, ends))
                    self._region_depth -= 1
                    if self._region_depth == 0 and self._deferred:
                        ready, self._deferred = self._deferred, []
                        for p in ready:         # (on the main stream, which now follows every branch of the region)
                            self.on_param_grad(p)
                    if self._region_depth == 0 and self.on_quiet is not None and tape:
                        self.on_quiet()
                continue
            if tag is None:
                fn()
                if self._region_depth == 0 and self.on_quiet is not None and tape:
                    self.on_quiet()             # (graph.GraphedTrainStep cuts its capture here when enough buckets are complete)
            else:
                with torch.cuda.stream(tag):
                    fn()
        self.grads.clear()
        self.shared.clear()
        self.bn_src.clear()
        self.bn_pre.clear()


# --------------------------------------------------------------------------- fused layers
def is_nhwc4(x):
    """network input already in the stem layout [B, H, W, 4] (utils.GpuIngest(..., nhwc4=True)) instead of NCHW [B, 3, H, W]"""
    return x.dim() == 4 and x.shape[-1] == 4 and x.shape[1] != 3


def image_hw(x):
    return tuple(x.shape[1:3]) if is_nhwc4(x) else tuple(x.shape[-2:])


TAPS = None   # diagnostic hook (tools/error_growth.py): a dict collects named intermediate activations (CPU copies, NCHW)


MARKS = None   # diagnostic hook (tools/stage_times.py): a list collects (direction, name, HIP event) at every tap of a pass and of its backward
_cur_cx = None  # the Ctx of the forward pass being recorded (EngineNet._run)


def tap(name, t):
    if MARKS is not None and t.is_cuda:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        MARKS.append(("fwd", name, ev))
        if _cur_cx is not None and _cur_cx.record:
            def mark():
                e2 = torch.cuda.Event(enable_timing=True)
                e2.record()
                MARKS.append(("bwd", name, e2))
            _cur_cx.push(mark)
    if TAPS is not None:
        if getattr(t, "_planes_only", False):       # (exists as fp16 x 2 planes only: h + l, rescaled)
            pl = t._planes
            B, H, W, C = pl.shape
            hl = pl.buf.view(torch.float16).view(2, C // 8, B * H * W, 8).float()
            v = (hl[0] + hl[1]).permute(1, 0, 2).reshape(B, H, W, C) * 2.0 ** (-int(pl.rec[1]))
            TAPS[name] = v.permute(0, 3, 1, 2).contiguous().cpu()
        else:
            TAPS[name] = t.detach().permute(0, 3, 1, 2).contiguous().cpu()
    return t


FUSE_BN_STATS = True  # training forward: BatchNorm batch statistics from the convolution epilogue (False: separate statistics pass)
FUSE_EVAL_BN = True   # eval-mode forward: fold BatchNorm into the conv and fuse bias/residual/ReLU into its epilogue


def _fused_head(cx, x, x_in, y, stats, scale, conv, bn, head, need_dx, pad_to):
    """BatchNorm + ReLU + the K-class 1 x 1 classifier `head` behind the head convolution `conv` whose output y (and batch statistics) exist:
    the normalised activation and its gradient are never written (ops.head_fwd / ops.head_backward, csrc/headfuse.h); the convolution's
    backward streams the blocked planes of dy as on conv_bn_act's head route."""
    w, hw = conv.weight, head.weight
    Cout, K = w.shape[0], hw.shape[0]
    kh, kw = conv.kernel_size
    s, p, d = conv.stride[0], conv.padding[0], conv.dilation[0]
    cx.claim(hw, head.bias)
    hb = head.bias.data if head.bias is not None else None
    logits = ops.head_fwd(y, stats[:Cout], scale, bn.bias.data, hw.data, hb, K, max(pad_to, (K + 3) // 4 * 4))

    def bwd():
        dl = cx.take(logits)
        if dl is None:
            return
        dyp, dysc = ops.head_backward(dl, y, stats, bn.weight.data, bn.bias.data, hw.data, cx.pgrad(hw),
                                      cx.pgrad(head.bias) if head.bias is not None else None, cx.pgrad(bn.weight), cx.pgrad(bn.bias),
                                      cx.pgrad(conv.bias) if conv.bias is not None else None)
        del dl
        cx.done(hw, head.bias)
        ops.conv_bwd_weight_h2(x_in, dyp, dysc, Cout, cx.pgrad(w), kh, kw, s, p, d)
        if need_dx:
            dx, accx = cx.dest(x)
            ops.conv_bwd_data_h2(dyp, dysc, w.data, tuple(x.shape), Cout, kh, kw, p, d, dx, accx)
        cx.done(bn.weight, bn.bias, w, conv.bias)
    cx.push(bwd)
    return logits


def conv_bn_act(cx, x, conv, bn, relu=True, residual=None, out=None, need_dx=True, private_in=False, sole_conv_out=False, head=None, z_tap=None):
    """conv -> BatchNorm (batch stats in training) -> (+residual) -> (ReLU).  x NHWC (or the raw
    NCHW image for the stem).  Returns z (NHWC).
    head: a 1 x 1 classifier convolution that is the ONLY consumer of z -- the call then returns conv_bias(z, head), and in a recorded training
    pass on the head layers' route BatchNorm, ReLU and classifier run fused (_fused_head: z is never written); z_tap names z for the diagnostic
    taps where it exists.
    private_in: the caller states that x is the output of the preceding conv_bn_act (ReLU, no residual) and has NO other consumer
    (the first half of a BasicBlock).  The backward-data kernel of this layer may then run the first pass of that BatchNorm's
    backward in its epilogue (ops.conv_bwd_data(bn_src=...)).
    sole_conv_out: the caller states that the OUTPUT of this layer feeds nothing but one conv_bn_act(private_in=True) (the same first half
    of a BasicBlock, seen from the producer): on the planes route (ops.planes_ok) it then exists as fp16 x 2 planes only."""
    w = conv.weight
    Cout = w.shape[0]
    kh, kw = conv.kernel_size
    s, p, d = conv.stride[0], conv.padding[0], conv.dilation[0]
    pad3 = (not conv.stem) and w.shape[1] == 3       # 3-channel image into a generic conv (HRNet 3x3/2 stem)
    stem3 = stem7 = False
    if (conv.stem and cx.train and conv.bias is None and not need_dx and out is None and w.data.is_contiguous(memory_format=torch.channels_last)
            and ops.stem7_ok(x, w.data, kh, kw, s, p, d, conv.groups)):
        # the ResNet stem's first convolution, training forward: the direct fp64-accumulating kernel on the image as it is (csrc/stem7.hip);
        # the backward-weight pass packs its operands for the implicit GEMM itself
        stem7 = True
        x_in, wk = x, w.data
    elif conv.stem:
        x_in = x if is_nhwc4(x) else ops.nchw3_to_nhwc4(x)
        wk = ops.stem_pack_weight(w.data, Cout)
    elif pad3 and cx.train and conv.bias is None and not need_dx and out is None and ops.stem3_ok(x, w.data, kh, kw, s, p, d, conv.groups):
        # the HRNet stem's first convolution: HBM-bound direct kernels on the image as it is (NCHW or NHWC-4), no repack, no channel padding
        stem3 = True
        x_in, wk = x, w.data
    elif pad3:
        x_in = x if is_nhwc4(x) else ops.nchw3_to_nhwc4(x)
        wk = ops.weight_pad_cin(w.data, Cout, kh * kw, 3, 4)
    else:
        x_in, wk = x, w.data
    bias = conv.bias.data if conv.bias is not None else None
    cx.claim(w, conv.bias, bn.weight, bn.bias)
    zrec = yrec = None
    if not cx.train and not cx.record and FUSE_EVAL_BN:
        # inference fast path: eval-mode BatchNorm folded into the weights, bias + residual + ReLU applied in
        # the convolution epilogue — one kernel per layer, no separate normalisation pass over HBM
        per_out = wk.numel() // Cout
        wf, bf = ops.fold_bn(wk, bias, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.eps, Cout, per_out)
        zf = ops.conv_fwd_fused(x_in, wf, bf, residual, relu, Cout, kh, kw, s, p, d, out=out, stem4=conv.stem, groups=conv.groups)
        if z_tap is not None:
            tap(z_tap, zf)
        return zf if head is None else conv_bias(cx, zf, head)
    if cx.train:
        # batch statistics: per-tile partial sums come out of the convolution's epilogue (no separate pass over y)
        if stem3:
            y = ops.stem3_fwd(x_in, wk, None, bn_stats=FUSE_BN_STATS)
        elif stem7:
            y = ops.stem7_fwd(x_in, wk, None, bn_stats=FUSE_BN_STATS)
        else:
            y = ops.conv_fwd(x_in, wk, bias, Cout, kh, kw, s, p, d, stem4=conv.stem, groups=conv.groups, bn_stats=FUSE_BN_STATS, train=cx.record,
                             exact=conv.exact_operands)
        y, partials = y if FUSE_BN_STATS else (y, None)
        # planes route: the convolution streamed planes and left max|y|; then the BatchNorm's output gets planes too (exponent from a bound)
        yrec = getattr(y, "_yrec", None)
        if (yrec is not None and partials is not None and out is None and ops.planes_ok(Cout, ops.rows_of(y))
                and (residual is None or ops.amax_of(residual) is not None)):
            zrec = ops.new_amax(y.device)
        if zrec is not None:
            stats, scale = ops.bn_finalize(partials, ops.rows_of(y), Cout, bn.weight.data, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                                           bound=(bn.bias.data, yrec, zrec))
        elif partials is not None:
            stats, scale = ops.bn_finalize(partials, ops.rows_of(y), Cout, bn.weight.data, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        else:
            stats, scale = ops.bn_train_stats(y, bn.weight.data, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        bn._pending_batches += 1
        mean = stats[:Cout]
    else:
        y = ops.conv_fwd(x_in, wk, bias, Cout, kh, kw, s, p, d, stem4=conv.stem, groups=conv.groups, train=cx.record)
        stats = None
        mean = bn.running_mean
        scale = ops.bn_eval_scale(bn.weight.data, bn.running_var, bn.eps)
    if (head is not None and cx.train and cx.record and TAPS is None and zrec is None and residual is None and relu and out is None
            and not conv.stem and not pad3 and partials is not None and head.kernel_size == (1, 1) and head.stride == (1, 1)
            and head.padding == (0, 0) and head.groups == 1 and ops.head_fuse_ok(y, head.weight.shape[0])
            and ops.h2_dy_route(x_in, y, w.data, kh, kw, s, p, d, conv.groups, need_dx)):
        return _fused_head(cx, x, x_in, y, stats, scale, conv, bn, head, need_dx, 32)
    z = ops.bn_apply(y, mean, scale, bn.bias.data, residual, relu, out=out, planes_rec=zrec,
                     planes_only=zrec is not None and sole_conv_out and relu and residual is None,
                     want_mask=cx.record and cx.train)       # (a residual block's output: its ReLU mask as bits for the backward pass)
    if cx.record:
        if not cx.train:
            raise NotImplementedError("backward through eval-mode BatchNorm is not on the training path")

        if relu and residual is None and out is None:
            cx.bn_src[id(z)] = (y, stats, bn.weight.data, bn.bias.data)     # for a private_in consumer of z

        def bwd():
            dz = cx.take(z)
            pre = cx.bn_pre.pop(id(z), None)
            cx.bn_src.pop(id(z), None)          # (its private_in consumer, if any, has run: y is not kept alive beyond this layer's backward)
            if dz is None:
                return
            dres, acc = (None, False)
            if residual is not None:
                dres, acc = cx.dest(residual)
            if yrec is not None and (pre is None or len(pre) == 3) and ops.planes_of(x_in) is not None and conv.bias is None:
                # planes route: the gradient of the convolution's output exists as planes only; backward-weight and backward-data stream them
                if pre is not None:
                    dyp = ops.bn_backward_pre_planes(dz, y, stats, bn.weight.data, pre, yrec, cx.pgrad(bn.weight), cx.pgrad(bn.bias))
                else:
                    z_mask = z if (residual is not None or not relu) else None
                    dyp = ops.bn_backward_planes(dz, z_mask, y, stats, bn.weight.data, relu, cx.pgrad(bn.weight), cx.pgrad(bn.bias), dres, acc,
                                                 bn.bias.data, yrec)
                del dz
                ops.dwgrad3_pl(ops.planes_of(x_in), dyp, cx.pgrad(w))
                if need_dx:
                    dx, accx = cx.dest(x)
                    src = cx.bn_src.get(id(x)) if (private_in and not accx) else None
                    r = ops.conv_bwd_data_pl(dyp, w.data, dx, accumulate=accx, bn_src=src)
                    if isinstance(r, tuple):
                        cx.bn_pre[id(x)] = r[1]
                cx.done(bn.weight, bn.bias, w, conv.bias)
                return
            if (pre is None and residual is None and not conv.stem and not pad3
                    and ops.h2_dy_route(x_in, y, w.data, kh, kw, s, p, d, conv.groups, need_dx)):
                # head layers on the f16x2 kernels: dy exists only as the blocked planes both of its consumers read (ops.bn_backward_h2)
                dyp, dysc = ops.bn_backward_h2(dz, y, stats, bn.weight.data, relu, cx.pgrad(bn.weight), cx.pgrad(bn.bias), bn.bias.data,
                                               cx.pgrad(conv.bias) if conv.bias is not None else None)
                del dz
                ops.conv_bwd_weight_h2(x_in, dyp, dysc, Cout, cx.pgrad(w), kh, kw, s, p, d)
                if need_dx:
                    dx, accx = cx.dest(x)
                    ops.conv_bwd_data_h2(dyp, dysc, w.data, tuple(x.shape), Cout, kh, kw, p, d, dx, accx)
                cx.done(bn.weight, bn.bias, w, conv.bias)
                return
            if pre is not None:
                # dz is already masked and its per-tile sums exist (the consumer's backward-data epilogue): merge + apply only
                dy = ops.bn_backward_pre(dz, y, stats, bn.weight.data, pre, cx.pgrad(bn.weight), cx.pgrad(bn.bias))
            else:
                # without a residual branch the ReLU mask is recomputed from y (z is not read: 1 of 3 tensor reads saved)
                z_mask = z if (residual is not None or not relu) else None
                dy = ops.bn_backward(dz, z_mask, y, stats, bn.weight.data, relu, cx.pgrad(bn.weight), cx.pgrad(bn.bias), dres, acc,
                                     beta=bn.bias.data)
            del dz
            dbias = cx.pgrad(conv.bias) if conv.bias is not None else None
            if conv.stem:
                x4 = x_in if is_nhwc4(x_in) else ops.nchw3_to_nhwc4(x_in)      # (stem7: the forward read the image as it was)
                dpk = torch.empty((Cout, 7, 8, 4), dtype=torch.float32, device=dy.device)     # the packed layout of ops.stem_pack_weight
                ops.conv_bwd_weight(x4, dy, dpk, dbias, kh, kw, s, p, d, stem4=True)
                ops.stem_unpack_grad(dpk, cx.pgrad(w), Cout)
            elif stem3:
                ops.stem3_bwd_weight(x_in, dy, cx.pgrad(w))
            elif pad3:
                dpk = torch.empty_like(wk)
                ops.conv_bwd_weight(x_in, dy, dpk, dbias, kh, kw, s, p, d)
                ops.weight_unpad_cin(dpk, cx.pgrad(w), Cout, kh * kw, 3, 4)
            else:
                ops.conv_bwd_weight(x_in, dy, cx.pgrad(w), dbias, kh, kw, s, p, d, groups=conv.groups)
                if need_dx:
                    dx, accx = cx.dest(x)
                    src = cx.bn_src.get(id(x)) if (private_in and not accx) else None
                    r = ops.conv_bwd_data(dy, w.data, tuple(x.shape), kh, kw, s, p, d, out=dx, accumulate=accx, groups=conv.groups, bn_src=src)
                    if isinstance(r, tuple):
                        cx.bn_pre[id(x)] = r[1]
            cx.done(bn.weight, bn.bias, w, conv.bias)
        cx.push(bwd)
    if z_tap is not None:
        tap(z_tap, z)
    return z if head is None else conv_bias(cx, z, head)


def conv_bias(cx, x, conv, pad_to=32):
    """plain conv (+bias), used for the K-class classifier heads.  The output keeps a zero-padded
    row stride (pad_to floats) so that it can feed 16-byte-granular kernels."""
    w = conv.weight
    Cout = w.shape[0]
    kh, kw = conv.kernel_size
    s, p, d = conv.stride[0], conv.padding[0], conv.dilation[0]
    ld = max(pad_to, (Cout + 3) // 4 * 4)
    bias = conv.bias.data if conv.bias is not None else None
    y = ops.conv_fwd(x, w.data, bias, Cout, kh, kw, s, p, d, zero_to=ld, train=cx.record)
    cx.claim(w, conv.bias)
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is None:
                return
            ops.conv_bwd_weight(x, dy, cx.pgrad(w), cx.pgrad(conv.bias) if conv.bias is not None else None, kh, kw, s, p, d)
            dx, acc = cx.dest(x)
            ops.conv_bwd_data(dy, w.data, tuple(x.shape), kh, kw, s, p, d, out=dx, accumulate=acc)
            cx.done(w, conv.bias)
        cx.push(bwd)
    return y


def conv_act(cx, x, conv, relu=True):
    """conv + bias (+ ReLU) without normalisation -- the VGG-style layers of models/FCN.py:42-55 of the reference: bias and ReLU run in
    the convolution's epilogue.  x NHWC (or the raw NCHW image for a 3-channel first layer)."""
    w = conv.weight
    Cout = w.shape[0]
    kh, kw = conv.kernel_size
    s, p, d = conv.stride[0], conv.padding[0], conv.dilation[0]
    pad3 = w.shape[1] == 3
    if pad3:
        x_in = x if is_nhwc4(x) else ops.nchw3_to_nhwc4(x)
        wk = ops.weight_pad_cin(w.data, Cout, kh * kw, 3, 4)
    else:
        x_in, wk = x, w.data
    bias = conv.bias.data if conv.bias is not None else None
    cx.claim(w, conv.bias)
    z = ops.conv_fwd_fused(x_in, wk, bias, None, relu, Cout, kh, kw, s, p, d)
    if cx.record:
        def bwd():
            dz = cx.take(z)
            if dz is None:
                return
            dy = ops.relu_bwd(dz, z) if relu else dz
            del dz
            dbias = cx.pgrad(conv.bias) if conv.bias is not None else None
            if pad3:
                dpk = torch.empty_like(wk)
                ops.conv_bwd_weight(x_in, dy, dpk, dbias, kh, kw, s, p, d)
                ops.weight_unpad_cin(dpk, cx.pgrad(w), Cout, kh * kw, 3, 4)
            else:
                ops.conv_bwd_weight(x_in, dy, cx.pgrad(w), dbias, kh, kw, s, p, d)
                dx, acc = cx.dest(x)
                ops.conv_bwd_data(dy, w.data, tuple(x.shape), kh, kw, s, p, d, out=dx, accumulate=acc)
            cx.done(w, conv.bias)
        cx.push(bwd)
    return z


def maxpool2(cx, x):
    """F.max_pool2d(x, 2) (models/FCN.py:44-53 of the reference)"""
    y, idx = ops.maxpool2_fwd(x)
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is None:
                return
            dx, acc = cx.dest(x)
            ops.maxpool2_bwd(dy, idx, dx, acc)
        cx.push(bwd)
    return y


class ConvTranspose2d(nn.ConvTranspose2d):
    """Parameter container with nn.ConvTranspose2d's init / state-dict behaviour (weight [Cin, Cout, k, k]); runs on the HIP engine."""

    def forward(self, x, output_size=None):  # pragma: no cover
        raise RuntimeError("engine ConvTranspose2d is executed by the owning network, not called directly")


def conv_transpose(cx, x, deconv):
    """nn.ConvTranspose2d on a class-logit tensor (rows zero padded to 32 floats: conv_bias's output).  The transposed convolution IS the
    backward-data pass of the convolution with the same weight tensor; its own backward is that convolution's forward / backward-weight."""
    w = deconv.weight
    Cin, Cout = w.shape[0], w.shape[1]
    k, s, p = deconv.kernel_size[0], deconv.stride[0], deconv.padding[0]
    assert deconv.kernel_size[0] == deconv.kernel_size[1] and deconv.output_padding[0] == 0 and deconv.dilation[0] == 1 and deconv.groups == 1
    bias = deconv.bias.data if deconv.bias is not None else None
    cx.claim(w, deconv.bias)
    y, wp = ops.conv_transpose_fwd(x, w.data, bias, Cout, k, s, p)
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is None:
                return
            dx, acc = cx.dest(x)
            ops.conv_transpose_bwd(dy, x, wp, cx.pgrad(w), cx.pgrad(deconv.bias) if deconv.bias is not None else None, k, s, p, dx, acc)
            cx.done(w, deconv.bias)
        cx.push(bwd)
    return y


def add_classes(cx, a, b):
    """a + b of two class-logit tensors (models/FCN.py:57,60 of the reference); rows zero padded to the same stride"""
    K = a.shape[-1]
    yw = ops.add_n_act([ops.widen(a), ops.widen(b)], False)
    y = yw[..., :K]
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is not None:
                cx.give(a, dy, shared=True)
                cx.give(b, dy, shared=True)
        cx.push(bwd)
    return y


def add_n(cx, terms, relu=True):
    """y = act(sum(terms)) — the HRNet fuse (models/HRNetv2.py:237-261); terms share one shape"""
    y = ops.add_n_act(terms, relu)
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is None:
                return
            g = ops.relu_bwd(dy, y) if relu else dy
            for t in terms:
                cx.give(t, g, shared=len(terms) > 1)
        cx.push(bwd)
    return y


def maxpool(cx, x):
    y, idx = ops.maxpool_fwd(x)
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is not None:
                cx.give(x, ops.maxpool_bwd(dy, idx, tuple(x.shape)))
        cx.push(bwd)
    return y


def bilinear_backward_of(cx, x, y, align_corners):
    """tape entry of y = bilinear(x): the gradient of y, resized back, goes to x"""
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is None:
                return
            dx, acc = cx.dest(x)
            C = x.shape[-1]
            ops.bilinear_bwd(dy, tuple(x.shape), align_corners, out=dx, zero_to=(ops.ld_of(dx) if ops.ld_of(dx) != C else 0),
                             accumulate=acc)
        cx.push(bwd)


def bilinear(cx, x, Ho, Wo, align_corners, out=None):
    y = ops.bilinear_fwd(x, Ho, Wo, align_corners, out=out)
    bilinear_backward_of(cx, x, y, align_corners)
    return y


def adaptive_avgpool(cx, x, S):
    """nn.AdaptiveAvgPool2d(S) -> [B, S, S, C]"""
    y = ops.adaptive_avgpool_fwd(x, S)
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is None:
                return
            dx, acc = cx.dest(x)
            ops.adaptive_avgpool_bwd(dy, dx, S, acc)
        cx.push(bwd)
    return y


def copy_into(cx, src, dst):
    """dst (a channel slice of a concat buffer) = src; the gradient of dst flows back to src"""
    rec = ops.amax_of(src)
    ops.axpy(src, dst, 1.0, False)
    if rec is not None:
        dst._amax = rec                 # (a copy: the source's record bounds it)
    if cx.record:
        def bwd():
            g = cx.take(dst)
            if g is not None:
                cx.give(src, g)
        cx.push(bwd)
    return dst


def global_avgpool(cx, x):
    y = ops.global_avgpool_fwd(x)
    if cx.record:
        def bwd():
            dy = cx.take(y)
            if dy is None:
                return
            dx, acc = cx.dest(x)
            ops.global_avgpool_bwd(dy, dx, acc)
        cx.push(bwd)
    return y


def concat_views(cx, cat, parts):
    """parts: [(tensor written as a channel slice of `cat`, c0, c1)] — backward hands out slices."""
    recs = [ops.amax_of(t) for t, _, _ in parts]
    if recs and all(r is not None for r in recs) and sum(c1 - c0 for _, c0, c1 in parts) == cat.shape[-1]:
        cat._amax_parts = recs          # every channel of the buffer has a producer that left max|.|: the f16x2 split pass needs no amax pass
    if cx.record:
        def bwd():
            dcat = cx.take(cat)
            if dcat is None:
                return
            for t, c0, c1 in parts:
                cx.give(t, dcat[..., c0:c1])
        cx.push(bwd)
    return cat


def spatial_gather(cx, feats, logits, K):
    """models/OCR.py:158-170: proxy[b, k, :] = sum_n softmax_n(logits[b, n, k]) * feats[b, n, :]"""
    B, H, W, C = feats.shape
    N = H * W
    ldl = ops.ld_of(logits)
    lbuf = torch.as_strided(logits, (B, N, ldl), (N * ldl, ldl, 1))
    probs = ops.softmax_spatial_fwd(lbuf, K)
    proxy = torch.empty((B, K, 1, C), dtype=torch.float32, device=feats.device)
    ldf = ops.ld_of(feats)
    ops.gemm_tn_split(B, K, C, N, probs, ldl, feats, ldf, proxy)
    if cx.record:
        def bwd():
            dproxy = cx.take(proxy)
            if dproxy is None:
                return
            dfe, acc = cx.dest(feats)
            ops.gemm(ops.NN, B, N, C, K, probs, ldl, N * ldl, dproxy, C, K * C, dfe, ops.ld_of(dfe), N * ops.ld_of(dfe),
                     accumulate=acc)
            dprobs = torch.empty_like(probs)
            ops.gemm(ops.NT, B, N, K, C, feats, ldf, N * ldf, dproxy, C, K * C, dprobs, ldl, N * ldl, zero_to=ldl)
            dlg, accl = cx.dest(logits)
            dbuf = torch.as_strided(dlg, (B, N, ldl), (N * ldl, ldl, 1))
            ops.softmax_spatial_bwd(probs, dprobs, dbuf, K, accumulate=accl)
        cx.push(bwd)
    return proxy


def object_attention_core(cx, q, key, val, K, key_channels):
    """models/OCR.py:266-274: softmax_k(C^-0.5 q.key) . val   (q [B,H,W,Ck]; key, val [B,K,1,Ck])"""
    B, H, W, Ck = q.shape
    N = H * W
    ld = 32 if K <= 32 else 64
    scale = float(key_channels) ** -0.5
    sim = torch.empty((B, N, ld), dtype=torch.float32, device=q.device)
    ops.gemm(ops.NT, B, N, K, Ck, q, Ck, N * Ck, key, Ck, K * Ck, sim, ld, N * ld, zero_to=ld)
    p = ops.softmax_rows_fwd(sim.view(B * N, ld), K, scale)
    del sim
    ctx = torch.empty((B, H, W, Ck), dtype=torch.float32, device=q.device)
    ops.gemm(ops.NN, B, N, Ck, K, p, ld, N * ld, val, Ck, K * Ck, ctx, Ck, N * Ck)
    if ops.amax_of(val) is not None:
        ctx._amax = ops.amax_of(val)    # every row of ctx is a convex combination of val's rows (softmax weights): max|ctx| <= max|val|
    if cx.record:
        def bwd():
            dctx = cx.take(ctx)
            if dctx is None:
                return
            dp = torch.empty_like(p)
            ops.gemm(ops.NT, B, N, K, Ck, dctx, Ck, N * Ck, val, Ck, K * Ck, dp, ld, N * ld, zero_to=ld)
            dv, accv = cx.dest(val)
            ops.gemm_tn_split(B, K, Ck, N, p, ld, dctx, Ck, dv, accumulate=accv)
            dsim = ops.softmax_rows_bwd(p, dp, K, scale)
            dq, accq = cx.dest(q)
            ops.gemm(ops.NN, B, N, Ck, K, dsim, ld, N * ld, key, Ck, K * Ck, dq, Ck, N * Ck, accumulate=accq)
            dk, acck = cx.dest(key)
            ops.gemm_tn_split(B, K, Ck, N, dsim, ld, q, Ck, dk, accumulate=acck)
        cx.push(bwd)
    return ctx


# --------------------------------------------------------------------------- autograd bridge
class NetFunction(torch.autograd.Function):
    """One autograd node for a whole network.  Parameter gradients are written directly into the
    flat gradient buffer by the tape; autograd only carries the output gradients in."""

    @staticmethod
    def forward(ctx, net, x, anchor):
        cx, outs = net._run(x, record=True)
        ctx.cx = cx
        ctx.net = net
        ctx.outs_nhwc = outs
        if net._keep_pass:
            net._last_pass = (cx, outs)
        res = tuple(o.permute(0, 3, 1, 2) for o in outs)
        ctx.mark_non_differentiable()
        return res if len(res) > 1 else res[0]

    @staticmethod
    def backward(ctx, *gouts):
        ctx.net._tape_backward(ctx.cx, ctx.outs_nhwc, gouts)
        return None, None, None


class EngineNet(nn.Module):
    """Base class of the HIP-engine networks (keeps the reference's ``Model(config, experiment)``
    / ``model(x)`` surface).  Subclasses implement ``_body(cx, x_nchw) -> [NHWC outputs]``."""

    def __init__(self):
        super().__init__()
        self._flatp = None
        self._grad_sync = None  # optional data-parallel gradient reducer
        self._grads_pending = False
        self._keep_pass = False  # graph.GraphedTrainStep (segmented capture): keep the recorded pass for backward_from()
        self._last_pass = self._last_outputs = None

    def flat(self):
        if self._flatp is None:
            self._flatp = FlatParams(self)
        return self._flatp.ensure()

    def state_dict(self, *a, **k):
        flush_bn_counters(self)
        return super().state_dict(*a, **k)

    def _d3_bank(self):
        """weight images of the direct 3x3 kernels (ops.Dconv3Bank): every eligible layer of the network, one launch per step"""
        fp = self.flat()
        bank = getattr(self, "_d3bank", None)
        h2 = ops._trunk_h2()
        if bank is None or (bank is not False and (bank.flat is not fp.flat or bank.h2 != h2)):
            ws = []
            for m in self.modules():
                if (isinstance(m, Conv2d) and not m.stem and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
                        and m.dilation == (1, 1) and m.groups == 1 and m.in_channels == m.out_channels
                        and ops.lib.catseg_dconv3_supported(m.in_channels) and id(m.weight) in fp.offsets):
                    ws.append((m.weight.data, fp.offsets[id(m.weight)]))
            bank = ops.Dconv3Bank(fp.flat, ws, h2=h2) if ws else False
            self._d3bank = bank
        return bank

    def _p1_bank(self):
        """weight images of the pointwise (1 x 1) layers for csrc/pconv1.hip (ops.P1Bank): every eligible layer, two launches per step"""
        fp = self.flat()
        bank = getattr(self, "_p1bank", None)
        if bank is None or (bank is not False and bank.flat is not fp.flat):
            ws = []
            for m in self.modules():
                if not (isinstance(m, Conv2d) and not m.stem and m.groups == 1 and id(m.weight) in fp.offsets and m.in_channels % 8 == 0
                        and m.stride[0] == m.stride[1] and m.padding[0] == m.padding[1] and m.dilation[0] == m.dilation[1]):
                    continue
                O, I, (kh, kw), st, pd, dl = m.out_channels, m.in_channels, m.kernel_size, m.stride[0], m.padding[0], m.dilation[0]
                if (kh, kw) == (1, 1) and st == 1 and pd == 0:
                    if ops.lib.catseg_pconv1_supported(O, I) and ops.lib.catseg_pconv1_supported(I, O):
                        ws.append((m.weight.data, fp.offsets[id(m.weight)], 1, 0, 1))
                elif ops.G1 and not (kh == 3 and kw == 3 and st == 1 and pd == 1 and dl == 1 and I == O and ops.lib.catseg_dconv3_supported(I)):
                    d = ops._g1_desc(I, O, kh, kw, st, pd, dl)
                    if (ops.lib.catseg_gconv_supported(__import__("ctypes").byref(d))
                            and ops.lib.catseg_pconv1_supported(I, ((kh + st - 1) // st) * ((kw + st - 1) // st) * O)):
                        ws.append((m.weight.data, fp.offsets[id(m.weight)], st, pd, dl))
            bank = ops.P1Bank(fp.flat, ws) if ws else False
            self._p1bank = bank
        return bank

    def _run(self, x, record):
        ops.release_b3_cache()          # (planes left over from a recorded forward that never saw its backward)
        # The per-step weight images (direct 3x3 kernels: 459 MB written for OCRNet-HRNet-W48, 0.43 ms; pointwise / gather kernels) are not
        # needed before stage 2: their launches run on a side stream beside the stem and stage 1; the launch stream waits for them at the
        # first lookup of an image or in front of the first parallel region (ops.images_ready).  Only while the step is being RECORDED into a
        # hipGraph (replay: 109.1 / 109.2 / 109.7 -> 108.5 / 109.2 / 109.0 ms): in the eager launch loop one more live stream shifts the runtime's
        # round-robin of streams over its four hardware queues, two branch streams of the parallel regions then share a queue and the step
        # LOSES 3 - 5 ms (116.4 / 119.5 against 113.5 / 114.3 ms; per-branch region times 1.9 / 2.7 / 2.7 / 2.7 instead of 2.2 / 2.3 / 2.0 / 2.3 ms).
        # PREP_ASYNC = False: in line everywhere.
        banks = []
        if ops.DCONV3 and ops.PRECISION == "bf16x3" and self.training:
            banks.append(self._d3_bank())
        if ops.P1 and ops._trunk_h2() and self.training:
            banks.append(self._p1_bank())
        banks = [b for b in banks if b]
        ops.h2_weight_images_begin(self.__dict__.setdefault("_h2w_images", {}))
        heads = self.training and ops.H2W_BANK and ops._h2w_bank
        if banks or heads:
            if PREP_ASYNC and x.is_cuda and torch.cuda.is_current_stream_capturing():
                main = torch.cuda.current_stream(x.device)
                side = prep_stream(x.device)
                side.wait_stream(main)          # (behind the optimiser's update of the weights and the last readers of the old images)
                with torch.cuda.stream(side):
                    for b in banks:
                        b.refresh()
                    if heads:
                        ops.h2_weight_images_refresh()      # (the head layers' images: in line, in front of their GEMMs, in the launch loop)
                    ev = torch.cuda.Event()
                    ev.record(side)
                ops.images_pending(ev, main)
            else:
                for b in banks:
                    b.refresh()
        cx = Ctx(self.training, record, None)
        if ops._trunk_h2() and self.training:
            # the per-tensor amax records of this forward pass and of its backward: ONE chunk, zeroed here on the main stream, sized from
            # the previous pass of this network (+25 %: a rollover inside a parallel region would have to be ordered against its siblings)
            cx.amax_scope = ops.AmaxScope(x.device, max(ops.AMAX_SCOPE_RECORDS, int(1.25 * getattr(self, "_amax_records", 0)) + 64))
            self._amax_scope_live = cx.amax_scope
            ops.set_amax_scope(cx.amax_scope)
        global _cur_cx
        _cur_cx = cx
        try:
            outs = self._body(cx, x)
        finally:
            _cur_cx = None
            ops.images_ready()          # (a pass that looked no image up -- small maps stay on the fp32 kernels -- still joins the prep stream)
        if not record:
            ops.release_b3_cache()      # a recorded forward keeps its split planes for the backward-weight pass (_end_backward frees them)
        return cx, outs

    def zero_grad(self, set_to_none=True):
        """clears the flat gradient buffer (one memset; the .grad views stay bound to it)"""
        fp = self.flat()
        if fp.grad is not None:
            fp.grad.zero_()
        self._grads_pending = False

    def _begin_backward(self, cx):
        fp = self.flat()
        if self._grads_pending:
            detached = sum(1 for p in fp.params if p.grad is None)
            if detached == len(fp.params) or detached > 0:
                # optimiser.zero_grad(set_to_none=True) of a stock torch optimiser (torch.optim.Adam(model.parameters()) in the reference's
                # loop, managers/OCRNet_Manager.py:80-90) detached every .grad: that IS the zero_grad between two backward passes.  An
                # optimiser over a SUBSET of the parameters detaches only its own: the others' gradients are read by nobody (torch would
                # accumulate them unseen; the tape overwrites them), so a partly detached set counts as cleared as well.
                fp.grad.zero_()
                self._grads_pending = False
            elif not bool(fp.grad.any()):
                # zero_grad(set_to_none=False): the .grad views were zeroed in place, i.e. the flat buffer is all zero (checked on this
                # rare path only: one reduction + a host synchronisation)
                self._grads_pending = False
            else:
                raise RuntimeError("second backward() before zero_grad(): the HIP engine's tape overwrites parameter gradients "
                                   "(no accumulation over several backward passes); call optimiser.zero_grad() / model.zero_grad() "
                                   "between backward passes")
        fp.bind_grads()
        if getattr(cx, "amax_scope", None) is not None:
            ops.set_amax_scope(cx.amax_scope)           # (another forward pass may have run since this one)
        if self._grad_sync is not None:
            cx.on_param_grad = self._grad_sync.param_ready
            cx.on_quiet = getattr(self._grad_sync, "quiet_point", None)
            self._grad_sync.begin(fp)

    def _tape_backward(self, cx, outs_nhwc, gouts):
        """the backward pass of a recorded forward: hands the output gradients (NCHW, as autograd carries them) to the tape and pops it"""
        for o, g in zip(outs_nhwc, gouts):
            if g is None:
                continue
            gn = g.permute(0, 2, 3, 1)
            if not gn.is_contiguous():
                gn = gn.contiguous()
            cx.give(o, gn)
        self._begin_backward(cx)
        cx.backward()
        self._end_backward(cx)

    def backward_from(self, gouts):
        """runs the tape of the last recorded forward pass ON THE CALLING THREAD from the gradients of its outputs (one per output of
        forward(), None allowed) -- what NetFunction.backward does inside autograd's device thread.  graph.GraphedTrainStep drives the
        backward pass this way when it cuts the captured step into segments: a stream capture has to end on the thread that began it.
        Needs `_keep_pass` set before the forward."""
        if self._last_pass is None:
            raise RuntimeError("backward_from(): no recorded forward pass is kept (set model._keep_pass = True before the forward)")
        (cx, outs), self._last_pass, self._last_outputs = self._last_pass, None, None
        self._tape_backward(cx, outs, gouts)

    def _end_backward(self, cx):
        if getattr(cx, "amax_scope", None) is not None:
            self._amax_records = max(getattr(self, "_amax_records", 0), cx.amax_scope.used)
        ops.release_b3_cache()
        self._grads_pending = True
        if self._grad_sync is not None:
            self._grad_sync.finish()

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("the HIP engine only runs on an MI355X device tensor (no CPU fallback)")
        fp = self.flat()
        x = x.contiguous().float()
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in fp.params):
            res = NetFunction.apply(self, x, fp.params[0])
            if self._keep_pass:
                self._last_outputs = res if isinstance(res, tuple) else (res,)
            return res
        _, outs = self._run(x, record=False)
        res = tuple(o.permute(0, 3, 1, 2) for o in outs)
        return res if len(res) > 1 else res[0]
