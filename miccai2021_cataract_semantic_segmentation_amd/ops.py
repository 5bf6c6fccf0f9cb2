"""Tensor-level wrappers over the C ABI (include/catseg.h).

Activations are NHWC fp32 torch tensors of shape [B, H, W, C] whose last dim is
contiguous and whose pixel stride ``ld = t.stride(2)`` may exceed C (a view into
a wider concat buffer).  Conv weights are [O, I, kh, kw] tensors in
channels_last memory format (physical OHWI).  Nothing here touches autograd.
"""
import ctypes

import torch

from . import _lib
from . import plan as _plan
from ._lib import ConvDesc, check, lib, ptr, stream

_ws = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream): launches on one stream are ordered; parallel-branch regions
    (engine.Ctx.parallel) run on side streams, each with its own scratch"""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def release_workspaces():
    _ws.clear()


def new_act(B, H, W, C, device, ld=None, zero=False):
    """NHWC activation; ld (pixel stride) defaults to C rounded up to 4."""
    ld = ld or ((C + 3) // 4 * 4)
    buf = (torch.zeros if zero else torch.empty)((B, H, W, ld), dtype=torch.float32, device=device)
    return buf[..., :C] if ld != C else buf


def ld_of(t):
    """pixel stride of an NHWC tensor; the stride torch reports for a size-1 dimension is arbitrary, so it is
    taken from the innermost pixel dimension that has more than one entry"""
    if t.dim() < 2:
        return t.shape[-1]
    inner = 1
    for dim in range(t.dim() - 2, -1, -1):
        if t.shape[dim] > 1:
            return t.stride(dim) // inner
        inner *= t.shape[dim]
    return (t.shape[-1] + 3) // 4 * 4


def rows_of(t):
    n = 1
    for s in t.shape[:-1]:
        n *= s
    return n


def conv_out_size(n, k, s, p, d):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def make_desc(xshape, ldx, Cout, ldy, kh, kw, stride, pad, dil, stem4=False, groups=1):
    B, H, W, Cin = xshape
    return ConvDesc(B, H, W, Cin, conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil), Cout,
                    kh, kw, stride, pad, dil, ldx, ldy, 1 if stem4 else 0, groups)


PROFILE = None  # bench.py sets this to a list: every implicit-GEMM launch is bracketed by HIP events

# roctx ranges: CATSEG_ROCTX=1 brackets every timed C-ABI call with roctxRangePush(kind) / roctxRangePop(), so that a
# `rocprofv3 --marker-trace --kernel-trace` timeline shows which layer operation a kernel belongs to.  Off by default (two ctypes calls per launch).
_roctx = None
if __import__("os").environ.get("CATSEG_ROCTX", "0") == "1":      # (a profiling aid, not a route: roctx ranges around every C-ABI call)
    for _name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            _roctx = ctypes.CDLL(_name)
            _roctx.roctxRangePushA.argtypes = [ctypes.c_char_p]
            break
        except (OSError, AttributeError):
            _roctx = None


class _Timed:
    """HIP events on the launch stream around one C-ABI call (algorithmic FLOPs = 2*M*N*K)."""

    def __init__(self, kind, flops):
        self.kind, self.flops = kind, flops

    def __enter__(self):
        if _roctx is not None:
            _roctx.roctxRangePushA(self.kind.encode())
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append((self.kind, self.flops, self.e0, self.e1))
        if _roctx is not None:
            _roctx.roctxRangePop()


# Arithmetic of the large convolutions: "fp32" = v_mfma_f32_32x32x2_f32 (an exact fp32 FMA chain), "bf16x3" = every fp32
# operand split exactly into three bf16 planes, six bf16 MFMA partial products per block, fp32 accumulate (csrc/igemm_bf16x3.hip:
# error vs an fp64 reference equal to or below the fp32 kernel's, 1.45-1.6x its speed on the layers it takes).  Only forward /
# backward-data of dense 3x3-class convolutions with a long reduction and >= 192 output columns qualify; everything else runs
# the fp32 kernels.  CATSEG_PRECISION=fp32 selects the exact fp32 path everywhere.
import os as _os
PRECISION = _plan.get("precision")
_b3_cache = {"key": None, "x": None, "planar": None, "blk": None}
# thresholds of the layer selection (tests lower them to push small layers through the split-precision kernels)
B3_MIN_TAPS, B3_MIN_K, B3_MIN_N, B3_MIN_TILES = 2, 2048, 192, 192
B3_OPS = ("fwd", "dgrad", "wgrad")
B3_MIN_WGRAD_ROWS = 32768
_b3_cache_dy = {"key": None, "x": None, "planar": None, "blk": None}


# wide 1x1 layers (the 1024 -> 512 bottleneck of the OCR head at stride 4): both GEMM extents >= 512, their product >= 512 * 1024,
# >= 131072 pixels -- there the split pass (one read + 1.5 writes of the activation) is paid back (tools/bench_b3.py: forward
# 2.38 -> 1.54 + 0.57 ms, backward-data 2.42 -> 1.80 + 0.30 ms at 8 x 136 x 240); smaller 1x1 layers stay on the fp32 kernels
# Round 3, with two fp16 planes and three products: the 1x1 layers of a ResNet50's layer 4 at stride 8 (2048 <-> 512 on 8 x 68 x 120 = 65 280
# pixels) pay as well -- OCRNet-R50 step 98.8 -> 87.5 ms (tools/ab_1x1_rows.py); smaller extents (256) still do not (87.6 -> 88.5 ms)
B3_1X1_MIN_DIM = _plan.get("wide_1x1_min_dim")
B3_1X1_MIN_PROD, B3_1X1_MIN_ROWS = B3_1X1_MIN_DIM * 1024, _plan.get("wide_1x1_min_rows")


def _b3_wide_1x1(rows, ncols, taps, cred):
    return taps == 1 and min(cred, ncols) >= B3_1X1_MIN_DIM and cred * ncols >= B3_1X1_MIN_PROD and rows >= B3_1X1_MIN_ROWS


B3_INDEX_LIMIT = 1 << 31     # the planar bf16x3 kernels index their operand planes with 32-bit element offsets


def _b3_eligible(rows, ncols, taps, cred, stride_ok=True):
    if not (PRECISION == "bf16x3" and stride_ok and taps <= 32 and cred % 8 == 0):
        return False
    if rows * cred >= B3_INDEX_LIMIT or ncols * taps * cred >= B3_INDEX_LIMIT:      # oversized operands: the fp32 kernels (64-bit addressing)
        return False
    if (B3_MIN_TAPS <= taps and taps * cred >= B3_MIN_K and ncols >= B3_MIN_N
            and ((rows + 255) // 256) * ((ncols + 255) // 256) >= B3_MIN_TILES):
        return True
    return _b3_wide_1x1(rows, ncols, taps, cred)


# Layers that run the 256 x 256 register-pipelined kernel (> 192 output columns) read BLOCKED planes in forward / backward-data
# (csrc/igemm_bf16x3.hip: whole cache lines per LDS-DMA instruction, 190 -> 248 TFLOP/s-equivalent on the 3x3 720 -> 512 layer);
# the backward-weight kernel keeps the planar planes, written by the same split pass when a backward will follow.
B3_BLOCKED = True
B3_PLANE_LIMIT = (1 << 32) - 64     # bytes of the three blocked planes of one operand (one 32-bit-offset buffer resource); tests lower it


def _b3_blocked_ok(ncols, cred, a_rows, w_rows, taps):
    """cred: channels of the gathered operand (a multiple of 16); the three planes of an operand must stay below 4 GB (the
    kernel addresses them through one 32-bit-offset buffer resource)"""
    return (B3_BLOCKED and ncols > 192 and cred % 16 == 0 and 6 * a_rows * cred < B3_PLANE_LIMIT
            and 6 * w_rows * taps * cred < B3_PLANE_LIMIT)


# Arithmetic of the layers on the blocked 256 x 256 path (forward / backward-data of the head convolutions): "f16x2" = two fp16 planes
# per operand after a per-tensor power-of-two prescale, three MFMA products (csrc/igemm_f16x2.hip: 22 significant bits per operand,
# half the matrix work of bf16x3); "bf16x3" = three exact bf16 planes, six products.  CATSEG_HEADS selects.
HEADS = _plan.get("heads")


def _h2():
    return HEADS == "f16x2"


# Backward-weight of the f16x2 layers from the BLOCKED planes its forward / backward-data read (csrc/igemm_f16x2.hip: igemm_h2t_kernel's
# blocked mode): one plane set per tensor, the split passes write half as much.  CATSEG_H2T=planar: the separate planar set of round 3.
H2T_BLOCKED = _plan.get("h2t") != "planar"


# launch-shape knobs of the library's debug interface (include/catseg_debug.h) as plan fields, so that a whole bench process can run under
# another setting (tools/ab_env_bench.sh).  0 = the library's default.
for _field, _setter in _plan.LIBRARY_KNOBS.items():
    if _plan.get(_field):
        getattr(lib, _setter)(int(_plan.get(_field)))


def plan():
    """the active execution plan: every route / threshold switch of the host layer with its live value (plan.FIELDS documents them);
    bench.py prints it as `config.plan`"""
    return _plan.active()


def _split3_any(x, want, both):
    """(planar, blocked) planes of x, at least the wanted one; `both`: produce the two layouts in ONE pass over x"""
    if want == "blk" or both:
        blk, planar = split3_blocked(x, with_planar=(want == "planar" or both))
        return planar, blk
    return split3(x), None


def _cached(cache, x, key, want, both):
    """want: "planar" / "blk" (bf16 x 3 planes), or "h2" / "h2p" ((fp16 x 2 planes in the blocked / planar layout, device scale record));
    both: on a miss produce the sibling layout in the same pass over x"""
    hit = cache["key"] == key and cache["x"] is x
    if not hit:
        cache.update(key=key, x=x, planar=None, blk=None, h2=None, h2p=None)
    if cache.get(want) is None:
        _refuse_h2_only(x, "a split pass (%s planes)" % want)
        if want in ("h2", "h2p"):
            blk, pl, sc = split2h(x, blocked=(want == "h2" or both), planar=(want == "h2p" or both))
            if blk is not None:
                cache["h2"] = (blk, sc)
            if pl is not None:
                cache["h2p"] = (pl, sc)
        else:
            planar, blk = _split3_any(x, want, both and not hit)
            if planar is not None:
                cache["planar"] = planar
            if blk is not None:
                cache["blk"] = blk
    return cache[want]


_b3_kept = {}   # training step: planes of every split input, kept from the forward pass for its backward-weight pass


def _split3_cached(x, want="planar", both=False, keep=False):
    """three-plane split of an activation.  The last one is always kept (the two 720-channel head convolutions of OCRNet-HRNet and
    the ASPP branches read one tensor); keep=True (a recorded training forward) holds on to the planes until release_b3_cache(),
    so that the layer's backward-weight pass reads what its forward produced instead of splitting x again (the planes of all
    bf16x3 layers of a step: 5.5 GB for OCRNet-HRNet-W48 at bs 8, of 288 GB)"""
    # (per stream: since round 5 a branch output may be read by fuse chains on SEVERAL branch streams -- each splits for itself; planes made
    #  on one stream are never read on another without an ordering event)
    hp = getattr(x, "_h2_planes", None)
    if hp is not None:          # a tensor that exists only as blocked f16x2 planes (register_h2_planes): they travel with it
        if want == "h2":
            return hp
        _refuse_h2_only(x, "a split pass (%s planes)" % want)
    key = (x.data_ptr(), x._version, tuple(x.shape), ld_of(x), torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0)
    ent = _b3_kept.get(key)
    if ent is not None and ent["x"] is x:
        return _cached(ent, x, key, want, both)
    if keep:
        ent = {"key": None, "x": None, "planar": None, "blk": None, "h2": None, "h2p": None}
        _b3_kept[key] = ent
        return _cached(ent, x, key, want, both)
    return _cached(_b3_cache, x, key, want, both)


def _split3_cached_dy(dy, want="planar", both=False):
    """the dy planes of a layer are used twice in its backward (backward-weight, then backward-data)"""
    return _cached(_b3_cache_dy, dy, (dy.data_ptr(), tuple(dy.shape), ld_of(dy)), want, both)


def release_b3_cache():
    _b3_cache.update(key=None, x=None, planar=None, blk=None, h2=None, h2p=None)
    _b3_cache_dy.update(key=None, x=None, planar=None, blk=None, h2=None, h2p=None)
    _b3_kept.clear()
    _d3_wimg.clear()
    _p1_wimg.clear()
    del _p1_keep[:]


# Direct 3x3 / stride 1 / pad 1 convolution of the HRNet trunk widths in split precision (csrc/dconv3_b3.hip): the fp32 activation is
# split into bf16 planes inside the kernel, a block reads the halo tile of its pixel tile once.  DCONV3_MIN_ROWS: below that many
# pixels the launch cannot fill the chip with its 128-pixel tiles and the fp32 implicit GEMM is as good (tests lower it).
DCONV3 = True
DCONV3_MIN_ROWS = 2048
_d3_wimg = {}


def _d3_ok(rows, Cin, Cout, kh, kw, stride, pad, dil, groups):
    return (DCONV3 and PRECISION == "bf16x3" and kh == 3 and kw == 3 and stride == 1 and pad == 1 and dil == 1 and groups == 1
            and Cin == Cout and rows >= DCONV3_MIN_ROWS and lib.catseg_dconv3_supported(Cin))


# Arithmetic of the direct trunk kernels' forward / backward-data: "f16x2" = two fp16 planes, three products (csrc/dconv3_f16x2.hip) for
# every launch whose input carries an amax record (left by its producer: bn_apply, add_n_act, bn_backward), "bf16x3" = three bf16 planes,
# six products.  CATSEG_TRUNK selects; inputs without a record always take the bf16x3 kernel.
TRUNK = _plan.get("trunk")


def _trunk_h2():
    return TRUNK == "f16x2" and PRECISION == "bf16x3"


# The trunk on PRODUCER-WRITTEN planes (round 4; csrc/planes.h, dconv3_pl.hip, dwgrad3_pl.hip): BatchNorm apply / the HRNet fuse sum / BatchNorm
# backward write the fp16 x 2 operand planes of what they produce, the direct kernels stream them by LDS-DMA.  CATSEG_TRUNK_PLANES=0: the
# round-3 route (fp32 tensors + amax records, split inside the convolution kernels).
PLANES = _plan.get("trunk_planes")


def _trunk_planes():
    return PLANES and _trunk_h2() and DCONV3


# Widths that take the planes route.  Measured within one run on the HRNet-W48 step (tools/ab_planes.py, min / median of 4 rounds, ms): fp32
# tensors + in-kernel split 118.9 / 128.9, planes for all four widths 118.4 / 118.6, planes for 96 / 192 / 384 only 117.1 / 117.5, planes for 48
# only 120.8 / 121.0.  The 48-channel layers stay on the round-3 kernel: standalone the planes kernel is 5 us faster there too (47.8 against
# 52.7 us), but it owns its CUs (eight waves of 128 registers per block, two blocks: every register of the SIMDs), whereas the uniform
# four-wave kernel leaves room for the BatchNorm / reduction kernels of the other branch streams to run beside it.
PLANES_WIDTHS = _plan.get("planes_widths")


def planes_ok(C, rows):
    """a trunk tensor of C channels and `rows` pixels takes the planes route: both plane kernels support the width"""
    return bool(_trunk_planes() and C in PLANES_WIDTHS and rows >= DCONV3_MIN_ROWS and lib.catseg_dconv3_pl_supported(C)
                and lib.catseg_dwgrad3_pl_supported(C))


AMAX_SCOPE_RECORDS = 1024
AMAX_WORDS = 512        # int32 words per record (include/catseg.h: CATSEG_AMAX_RECORD_BYTES): 16 slots 128 bytes apart
_amax_scope = None


class AmaxScope:
    """the amax records of ONE forward pass and its backward: zeroed chunks of records, handed out in order.  A record lives as long as a
    tensor (or this scope) refers to its chunk -- a second forward pass (another network, a second micro-batch before the first backward)
    gets its own scope and cannot clear records that are still waiting for their backward.
    A chunk is zero-filled on the stream that is current when it is created (the first one at the stem, on the main stream; `records` sizes
    it for the whole pass when the owner knows the count of its previous pass).  Records are handed out on other streams too (HRNet's
    branch regions): the first request from a stream that may not be ordered behind the fill waits for the fill's event, so that a late
    memset can never erase an amax a sibling stream has already accumulated."""

    def __init__(self, device, records=AMAX_SCOPE_RECORDS):
        self.device, self.chunk, self.n, self.i = device, None, max(int(records), 16), 0
        self.used = 0
        self.i = self.n        # (first new() allocates)
        self.event, self.ordered = None, set()

    def new(self):
        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if self.i >= self.n:
            self.chunk, self.i = torch.zeros(AMAX_WORDS * self.n, dtype=torch.int32, device=self.device), 0
            if cur is not None:
                self.event = torch.cuda.Event()
                self.event.record(cur)
                self.ordered = {cur.cuda_stream}
        elif cur is not None and cur.cuda_stream not in self.ordered:
            cur.wait_event(self.event)
            self.ordered.add(cur.cuda_stream)
        i = self.i
        self.i = i + 1
        self.used += 1
        return self.chunk[AMAX_WORDS * i:AMAX_WORDS * (i + 1)]


def set_amax_scope(scope):
    """the scope new_amax() draws from (the engine sets it at the start of a recorded forward and of its backward)"""
    global _amax_scope
    _amax_scope = scope


def new_amax(device):
    """a zeroed amax record (int32[AMAX_WORDS]) from the current scope (a private scope per device outside the engine)"""
    global _amax_scope
    if _amax_scope is None or _amax_scope.device != device:
        _amax_scope = AmaxScope(device)
    return _amax_scope.new()


def amax_of(t):
    """the record a producing kernel attached to t, or None"""
    return getattr(t, "_amax", None)


SPLIT_BOUND = _plan.get("split_bound")    # f16x2 split passes take max|x| from the producers' records when they exist


def amax_records_of(t):
    """the amax records that bound |t|: its own, or those of the channel slices of a concatenation buffer (engine.concat_views); None if
    unknown or more than four"""
    r = getattr(t, "_amax", None)
    if r is not None:
        return [r]
    parts = getattr(t, "_amax_parts", None)
    return parts if parts and len(parts) <= 4 else None


def drop_amax(t):
    """a tensor that is about to be modified IN PLACE (an accumulating launch) loses the amax record its producer attached: the record
    would underestimate the new contents, and an underestimate overflows the fp16 planes of the trunk kernels (their prescale leaves one
    bit of headroom).  Without a record the consumer takes the three-plane bf16 kernel, which needs none."""
    if t is not None and getattr(t, "_amax", None) is not None:
        t._amax = None
    if t is not None and getattr(t, "_planes", None) is not None:
        t._planes = None           # (planes describe the old contents)
    if t is not None and getattr(t, "_amax_parts", None) is not None:
        t._amax_parts = None
    return t


_images_event = None      # (event, origin stream): the step's weight images are being written on a side stream (engine.EngineNet._run)


def images_pending(ev, origin):
    global _images_event
    _images_event = (ev, origin)


def images_ready():
    """the current stream waits for the weight-image launches of this step (once: every later stream forks from the origin stream)"""
    global _images_event
    if _images_event is not None:
        ev, origin = _images_event
        cur = torch.cuda.current_stream(origin.device)
        cur.wait_event(ev)
        if cur.cuda_stream == origin.cuda_stream:
            _images_event = None


def dconv3_weight_image(w, backward_data=False, h2=False):
    """pre-split weight image of the direct kernel (cached until release_b3_cache(): one per layer and direction per step);
    h2: the two-plane fp16 image and its scale record, (image, record)"""
    images_ready()
    key = (w.data_ptr(), bool(backward_data), bool(h2))
    img = _d3_wimg.get(key)
    if img is None:
        C = w.shape[0]
        if h2:       # (layers outside a Dconv3Bank: a one-layer bank over the weight tensor itself)
            Dconv3Bank(w, [(w, 0)], h2=True).refresh()
            return _d3_wimg[key]
        img = torch.empty(lib.catseg_dconv3_wimg_bytes(C), dtype=torch.uint8, device=w.device)
        check(lib.catseg_dconv3_prep(ptr(w), C, 1 if backward_data else 0, ptr(img), stream()))
        _d3_wimg[key] = img
    return img


class Dconv3Bank:
    """the weight images of every direct-kernel layer of one network, written by ONE launch per step
    (catseg_dconv3_prep_batch over the flat parameter buffer) instead of two small launches per layer.
    h2: the two-plane fp16 images of csrc/dconv3_f16x2.hip with their per-layer scale records (amax + image launch)"""

    def __init__(self, flat, weights, h2=False):
        """weights: [(parameter tensor (a view into flat), offset in floats)] of the eligible 3x3 layers"""
        import numpy as np
        self.flat = flat
        self.h2 = h2
        rec = np.zeros(2 * len(weights), dtype=[("w", "<i8"), ("img", "<i8"), ("C", "<i4"), ("KC", "<i4"), ("NT", "<i4"), ("dg", "<i4")])
        off = 0
        self.slices = []
        kc, nt = ctypes.c_int(0), ctypes.c_int(0)
        for i, (w, woff) in enumerate(weights):
            C = w.shape[0]
            lib.catseg_dconv3_layout(C, ctypes.byref(kc), ctypes.byref(nt))
            nbytes = lib.catseg_dconv3_f16x2_wimg_bytes(C) if h2 else lib.catseg_dconv3_wimg_bytes(C)
            for dg in (0, 1):
                rec[2 * i + dg] = (woff, off, C, kc.value, nt.value, dg)
                self.slices.append((w, dg, off, nbytes, 2 * i + dg))
                off += (nbytes + 255) // 256 * 256
        self.entries = torch.from_numpy(rec.view(np.uint8).copy()).to(flat.device)
        self.n = len(rec)
        self.images = torch.empty(max(off, 256), dtype=torch.uint8, device=flat.device)
        self.records = torch.zeros(2 * self.n, dtype=torch.int32, device=flat.device) if h2 else None

    def refresh(self):
        """(re)write every image from the current parameters and publish them to dconv3_weight_image's cache"""
        if self.h2:
            check(lib.catseg_dconv3_f16x2_prep_batch(ptr(self.flat), self.n, ptr(self.entries), ptr(self.images), ptr(self.records), stream()))
            for w, dg, off, nbytes, k in self.slices:
                _d3_wimg[(w.data_ptr(), bool(dg), True)] = (self.images[off:off + nbytes], self.records[2 * k:2 * k + 2])
            return
        check(lib.catseg_dconv3_prep_batch(ptr(self.flat), self.n, ptr(self.entries), ptr(self.images), stream()))
        for w, dg, off, nbytes, k in self.slices:
            _d3_wimg[(w.data_ptr(), bool(dg), False)] = self.images[off:off + nbytes]


# Pointwise (1 x 1, stride 1) convolutions in split precision with the split in registers (csrc/pconv1.hip): every dense 1 x 1 layer whose
# input carries an amax record and that is too small for the blocked-plane kernels (ops._b3_wide_1x1) -- the stage-1 bottlenecks, the
# object-attention block of the OCR head, the HRNet fuse layers.  CATSEG_P1=0: the fp32 MFMA kernels, as in round 4.
# Measured (tools/time_p1.py, four tensors in turn, fp32 MFMA kernel -> this route, us per launch at the bench size): stage-1 64 -> 256 forward
# 178 -> 138, backward-data 113 -> 85; 256 -> 64 129 -> 81 / 135 -> 119; f_pixel 512 -> 256 609 -> 264 / 597 -> 326; 256 -> 256 337 -> 191 / 308 -> 176;
# f_up 256 -> 512 643 -> 354 / 555 -> 252 (2.3 - 4.1 TB/s of algorithmic bytes).  The fuse layers of the low-resolution branches (8 160 ... 65 280
# pixels: 32 ... 255 blocks of 256 rows) are no faster or slower (0.55 - 1.2 x): P1_MIN_ROWS keeps them on the fp32 kernels.  Backward-weight
# (p1t_kernel: LDS-bound, both operands pass through the transposing reads) pays for the 256 / 512-channel layers only (619 -> 492, 343 -> 276 us;
# 64-channel layers 0.82 - 0.89 x): P1_WGRAD_MIN_DIM.  HRNet-W48 step (tools/ab_p1.sh, graph replay, two alternating rounds): 109.06 / 109.97 ms
# without, 108.24 / 108.09 with all three operations on every layer, 107.72 / 107.65 with forward + backward-data only.
P1 = _plan.get("p1")
P1_MIN_ROWS = _plan.get("p1_min_rows")
P1_WGRAD_MIN_DIM = _plan.get("p1_wgrad_min_dim")
P1_OPS = _plan.get("p1_ops")
_p1_wimg = {}


def _p1_ok(rows, Cin, Cout, kh, kw, stride, pad, dil, groups):
    return (P1 and _trunk_h2() and kh == 1 and kw == 1 and stride == 1 and pad == 0 and groups == 1 and rows >= P1_MIN_ROWS
            and not _b3_wide_1x1(rows, Cout, 1, Cin))


_P1_ENTRY = [("w", "<i8"), ("img", "<i8"), ("O", "<i4"), ("I", "<i4"), ("kh", "<i4"), ("kw", "<i4"), ("t", "<i4"), ("ky0", "<i4"), ("kys", "<i4"),
             ("nky", "<i4"), ("kx0", "<i4"), ("kxs", "<i4"), ("nkx", "<i4"), ("pad", "<i4")]      # csrc/pconv1.hip: P1Entry (64 bytes)


def _g1_desc(Cin, Cout, kh, kw, stride, pad, dil):
    """conv descriptor for the shape-independent queries of the gather launches (csrc/pconv1.hip: catseg_gconv_*)"""
    return ConvDesc(0, 0, 0, Cin, 0, 0, Cout, kh, kw, stride, pad, dil, Cin, Cout, 0, 1)


class P1Bank:
    """the weight images of every pointwise AND gather layer of one network for csrc/pconv1.hip -- per layer the forward image and the image(s)
    of backward-data (1 x 1: the transposed image; kh x kw / stride s: one image per input-pixel parity class) -- written by ONE amax + ONE
    image launch per step over the flat parameter buffer (catseg_pconv1_prep_batch)"""

    def __init__(self, flat, weights):
        """weights: [(parameter tensor [O, I, kh, kw] (a view into flat), offset in floats, stride, pad, dil)]"""
        import numpy as np
        self.flat = flat
        recs, self.slices, off = [], [], 0
        for w, woff, stride, pad, dil in weights:
            O, I, kh, kw = w.shape
            if kh == 1 and kw == 1 and stride == 1:
                for t in (0, 1):
                    nbytes = lib.catseg_pconv1_wimg_bytes(I if t else O, O if t else I)
                    self.slices.append(((w.data_ptr(), bool(t)), off, nbytes, len(recs)))
                    recs.append((woff, off, O, I, 1, 1, t, 0, 1, 1, 0, 1, 1, 0))
                    off += (nbytes + 255) // 256 * 256
                continue
            d = _g1_desc(I, O, kh, kw, stride, pad, dil)
            for bwd in (0, 1):
                nbytes = lib.catseg_gconv_wimg_bytes(ctypes.byref(d), bwd)
                buf = np.zeros(4, dtype=_P1_ENTRY)
                n = lib.catseg_gconv_entries(ctypes.byref(d), bwd, woff, off, buf.ctypes.data)
                assert n >= 1
                self.slices.append(((w.data_ptr(), "g", bool(bwd), stride, pad, dil), off, nbytes, len(recs)))
                recs.extend(tuple(int(v) for v in buf[i]) for i in range(n))
                off += (nbytes + 255) // 256 * 256
        rec = np.array(recs, dtype=_P1_ENTRY)
        self.entries = torch.from_numpy(rec.view(np.uint8).copy()).to(flat.device)
        self.n = len(rec)
        self.images = torch.empty(max(off, 256), dtype=torch.uint8, device=flat.device)
        self.records = torch.zeros(2 * self.n, dtype=torch.int32, device=flat.device)

    def refresh(self):
        check(lib.catseg_pconv1_prep_batch(ptr(self.flat), self.n, ptr(self.entries), ptr(self.images), ptr(self.records), stream()))
        for key, off, nbytes, k in self.slices:
            _p1_wimg[key] = (self.images[off:off + nbytes], self.records[2 * k:2 * k + 2])


def p1_weight_image(w, transposed=False):
    """(image, record) of a pointwise layer's weights [O, I, 1, 1] (physical OHWI = [O][I]); layers outside a P1Bank get a one-layer bank"""
    images_ready()
    key = (w.data_ptr(), bool(transposed))
    img = _p1_wimg.get(key)
    if img is None:
        bank = P1Bank(w, [(w, 0, 1, 0, 1)])
        bank.refresh()
        _p1_keep.append(bank)
        img = _p1_wimg[key]
    return img


# Gather launches of the same kernels (csrc/pconv1.hip: catseg_gconv_*): dense kh x kw convolutions with stride 1 / 2 whose input carries an
# amax record and that neither the direct 3x3 kernels nor the blocked-plane kernels take -- the stride-2 layers of the HRNet fuse chains and
# transitions, the 256 -> 48 transition, the stem's second convolution.  CATSEG_G1=0: the fp32 MFMA kernels.
# Measured (tools/time_g1.py, fp32 MFMA kernel -> gather launch, us): 256 -> 48 3x3 at 8 x 136 x 240: forward 585 -> 398, backward-data 495 -> 249,
# backward-weight 739 -> 651; 256 -> 96 / stride 2: 251 -> 129, 315 -> 200, 348 -> 230; stem 64 -> 64 / stride 2: 225 -> 143 forward; 48 -> 96 / stride 2:
# 72 -> 46 forward, 107 -> 73 backward-weight -- but its backward-data (four parity-class launches with N = 48 columns) 88 -> 95 and 48 -> 48: 59 -> 76:
# backward-data takes the route from G1_DGRAD_MIN_CIN input channels on.  HRNet-W48 step (graph replay, three alternating rounds): 108.1 - 108.8 ms
# without, 107.9 - 108.2 with.
G1 = _plan.get("g1")
G1_MIN_ROWS = _plan.get("g1_min_rows")
G1_DGRAD_MIN_CIN = _plan.get("g1_dgrad_min_cin")
G1_MIN_CIN = _plan.get("g1_min_cin")
G1_OPS = _plan.get("g1_ops")


def _g1_ok(rows, Cin, Cout, kh, kw, stride, pad, dil, groups):
    return (G1 and _trunk_h2() and groups == 1 and (kh * kw > 1 or stride > 1) and rows >= G1_MIN_ROWS and Cin % 8 == 0 and Cin >= G1_MIN_CIN
            and not _b3_eligible(rows, Cout, kh * kw, Cin, stride == 1))


def g1_weight_image(w, backward_data, stride, pad, dil):
    """(image(s), record) of a gather layer: the forward image, or the stride^2 class images of backward-data back to back"""
    images_ready()
    key = (w.data_ptr(), "g", bool(backward_data), stride, pad, dil)
    img = _p1_wimg.get(key)
    if img is None:
        bank = P1Bank(w, [(w, 0, stride, pad, dil)])
        bank.refresh()
        _p1_keep.append(bank)
        img = _p1_wimg[key]
    return img


_p1_keep = []


def pconv1(x, wimg, bias, N, out, accumulate=False, bn_stats=False):
    """out[rows][N] (+)= x[rows][K] . B^T (+ bias) through csrc/pconv1.hip; x carries an amax record; wimg = p1_weight_image(...)"""
    rows, K = rows_of(x), x.shape[-1]
    part = tr = nt = None
    if bn_stats:
        part = _bn_part_buffer(3 * ((rows + 255) // 256) * N, x.device)
        tr, nt = ctypes.c_int(0), ctypes.c_int(0)
    check(lib.catseg_pconv1(rows, N, K, ptr(x), ld_of(x), ptr(amax_of(x)), ptr(wimg[0]), ptr(wimg[1]), ptr(bias), ptr(out), ld_of(out),
                            1 if accumulate else 0, ptr(part), part.numel() if part is not None else 0,
                            ctypes.byref(tr) if bn_stats else None, ctypes.byref(nt) if bn_stats else None, stream()))
    if bn_stats:
        return out, ((part, nt.value, tr.value) if tr.value > 0 else None)
    return out


def dconv3(x, wimg, bias=None, out=None, accumulate=False, bn_stats=False, x_amax=None):
    """y (+)= conv3x3(x) from a weight image; bn_stats: also returns (partials, n_tiles, 0, counts) for bn_finalize.
    x_amax: the input's amax record -> the two-plane fp16 kernel; wimg is then (image, record) of dconv3_weight_image(..., h2=True)"""
    B, H, W, C = x.shape
    if out is None:
        out = new_act(B, H, W, C, x.device)
        accumulate = False
    drop_amax(out)
    part = cnt = None
    nt = 0
    if bn_stats:
        nt = lib.catseg_dconv3_tiles(C, B, H, W, None, None)
        part = _bn_part_buffer(3 * nt * C + nt, x.device)
        cnt = part[3 * nt * C:3 * nt * C + nt].view(torch.int32)
    if x_amax is not None:
        check(lib.catseg_dconv3_f16x2(B, H, W, C, ptr(x), ld_of(x), ptr(x_amax), ptr(wimg[0]), ptr(wimg[1]), ptr(bias), ptr(out), ld_of(out),
                                      1 if accumulate else 0, ptr(part), 3 * nt * C, ptr(cnt), stream()))
    else:
        check(lib.catseg_dconv3(B, H, W, C, ptr(x), ld_of(x), ptr(wimg), ptr(bias), ptr(out), ld_of(out), 1 if accumulate else 0,
                                ptr(part), 3 * nt * C, ptr(cnt), stream()))
    if bn_stats:
        return out, (part, nt, 0, cnt)
    return out


class Planes:
    """the fp16 x 2 operand planes of an NHWC activation (csrc/planes.h) + its record (amax slots, word 1 = the planes' exponent)"""
    __slots__ = ("buf", "rec", "shape")

    def __init__(self, buf, rec, shape):
        self.buf, self.rec, self.shape = buf, rec, tuple(shape)


def planes_of(t):
    """the planes a producing kernel attached to t, or None"""
    return getattr(t, "_planes", None)


def planes_from_f32(x, rec=None):
    """standalone producer: planes of x with the exponent from max|x| (rec given: its amax slots are taken as they are)"""
    B, H, W, C = x.shape
    buf = torch.empty(lib.catseg_planes_bytes(B * H * W, C), dtype=torch.uint8, device=x.device)
    own = rec is None
    if own:
        rec = new_amax(x.device)
    check(lib.catseg_planes_from_f32(ptr(x), ld_of(x), B * H * W, C, ptr(rec), ptr(buf), 1 if own else 0, stream()))
    return Planes(buf, rec, x.shape)


def dconv3_pl(xp, wimg, bias=None, out=None, accumulate=False, bn_stats=False, out_rec=None):
    """y (+)= conv3x3(x) from the planes of x (Planes) and an f16x2 weight image (image, record); bn_stats: also returns
    (partials, n_rows, 0, counts) for bn_finalize -- per-WAVE rows"""
    B, H, W, C = xp.shape
    if out is None:
        out = new_act(B, H, W, C, xp.buf.device)
        accumulate = False
    drop_amax(out)
    part = cnt = None
    nr = 0
    if bn_stats:
        nr = lib.catseg_dconv3_pl_rows(C, B, H, W)
        part = _bn_part_buffer(3 * nr * C + nr, xp.buf.device)
        cnt = part[3 * nr * C:3 * nr * C + nr].view(torch.int32)
    check(lib.catseg_dconv3_pl(B, H, W, C, ptr(xp.buf), ptr(xp.rec), ptr(wimg[0]), ptr(wimg[1]), ptr(bias), ptr(out), ld_of(out),
                               1 if accumulate else 0, ptr(part), 3 * nr * C, ptr(cnt), ptr(out_rec), stream()))
    if bn_stats:
        return out, (part, nr, 0, cnt)
    return out


def dwgrad3_pl(xp, dyp, dw):
    """backward-weight of a 3x3 / stride 1 / pad 1 trunk convolution from the planes of x and dy (Planes)"""
    B, H, W, C = xp.shape
    need = lib.catseg_dwgrad3_pl_workspace(B, H, W, C)
    ws = workspace(need + 256, xp.buf.device)
    with _Timed("wgrad_d3p", 2.0 * B * H * W * C * C * 9):
        check(lib.catseg_dwgrad3_pl(B, H, W, C, ptr(xp.buf), ptr(xp.rec), ptr(dyp.buf), ptr(dyp.rec), ptr(dw), ptr(ws), need, stream()))
    return dw


_bn_part = {}


def _bn_part_buffer(nfloats, device):
    """grow-only buffer (per device and stream) for the per-tile BatchNorm partials a convolution epilogue writes (consumed by
    the next launch on the same stream)"""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    buf = _bn_part.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = torch.empty(max(int(nfloats), 1 << 18), dtype=torch.float32, device=device)
        _bn_part[key] = buf
    return buf


def _refuse_placeholder(t, what):
    """a planes-only activation (bn_apply(planes_only=True)) is an UNWRITTEN fp32 placeholder: only the plane-streaming kernels may consume
    it.  Every other route of the convolution wrappers refuses it instead of reading uninitialised memory (a consumer with
    exact_operands, a failed _d3_ok, a bias, or CATSEG_PLANES_WIDTHS / DCONV3 toggled between producer and consumer would get here)."""
    if getattr(t, "_planes_only", False):
        raise RuntimeError("a planes-only activation reached %s, which reads fp32: its fp32 tensor was never written "
                           "(engine.conv_bn_act(sole_conv_out=True) requires the planes route on the consumer)" % what)


# The first stem convolution of HRNet (3 -> 64, 3 x 3 / stride 2 / pad 1 on the image) as HBM-bound direct fp32 kernels (csrc/stem3.hip) instead of
# the implicit GEMM over 9 taps x 4 padded channels.  CATSEG_STEM3=0: the implicit-GEMM route.
STEM3 = _plan.get("stem3")


def _image_strides(x):
    """(B, H, W, sb, sc, sy, sx) of a 3-channel image given as NCHW [B, 3, H, W] or NHWC-4 [B, H, W, 4]"""
    if x.dim() == 4 and x.shape[-1] == 4 and x.shape[1] != 3:
        B, H, W, _ = x.shape
        return B, H, W, x.stride(0), 1, x.stride(1), x.stride(2)
    B, _, H, W = x.shape
    return B, H, W, x.stride(0), x.stride(1), x.stride(2), x.stride(3)


def stem3_ok(x, w, kh, kw, stride, pad, dil, groups):
    if not (STEM3 and x.is_cuda and x.dtype == torch.float32 and w.dim() == 4 and w.shape[1] == 3 and (kh, kw, stride, pad, dil, groups) == (3, 3, 2, 1, 1, 1)):
        return False
    B, H, W = _image_strides(x)[:3]
    return bool(lib.catseg_stem3_supported(H, W, w.shape[0]))


def stem3_fwd(x, w, bias, bn_stats=False):
    """x: the image (NCHW or NHWC-4, any strides); w: [64, 3, 3, 3] in channels_last memory (physical OHWI).  Returns y NHWC [B, Ho, Wo, 64]
    (+ the BatchNorm partials (part, rows, 0, counts) for bn_finalize)"""
    B, H, W, sb, sc, sy, sx = _image_strides(x)
    Cout = w.shape[0]
    Ho, Wo = conv_out_size(H, 3, 2, 1, 1), conv_out_size(W, 3, 2, 1, 1)
    out = new_act(B, Ho, Wo, Cout, x.device)
    part = cnt = None
    nt = 0
    if bn_stats:
        nt = lib.catseg_stem3_partial_rows(B, H, W)
        part = _bn_part_buffer(3 * nt * Cout + nt, x.device)
        cnt = part[3 * nt * Cout:3 * nt * Cout + nt].view(torch.int32)
    with _Timed("hbm:stem3", 4.0 * (B * 3 * H * W + out.numel())):
        check(lib.catseg_stem3_fwd(ptr(x), sb, sc, sy, sx, B, H, W, ptr(w), ptr(bias), ptr(out), ld_of(out), ptr(part), ptr(cnt), stream()))
    return (out, (part, nt, 0, cnt)) if bn_stats else out


def stem3_bwd_weight(x, dy, dw):
    B, H, W, sb, sc, sy, sx = _image_strides(x)
    need = lib.catseg_stem3_wgrad_workspace()
    ws = workspace(need, x.device)
    with _Timed("hbm:stem3", 4.0 * (B * 3 * H * W + dy.numel())):
        check(lib.catseg_stem3_bwd_weight(ptr(x), sb, sc, sy, sx, B, H, W, ptr(dy), ld_of(dy), ptr(dw), ptr(ws), need, stream()))
    return dw


# The first convolution of the torchvision ResNet stem (3 -> 64, 7 x 7 / stride 2 / pad 3 on the image), training forward accumulated in fp64 and
# rounded once (csrc/stem7.hip): what csrc/stem3.hip did for the HRNet models' literal-1e-3 distance to the CPU path, for OCRNet-R50 / DeepLabv3+.
# The layer's backward-weight stays on the implicit GEMM (stem4 layout).  CATSEG_STEM7=0: the implicit-GEMM forward.
STEM7 = _plan.get("stem7")


def stem7_ok(x, w, kh, kw, stride, pad, dil, groups):
    if not (STEM7 and x.is_cuda and x.dtype == torch.float32 and w.dim() == 4 and w.shape[1] == 3 and (kh, kw, stride, pad, dil, groups) == (7, 7, 2, 3, 1, 1)):
        return False
    B, H, W = _image_strides(x)[:3]
    return bool(lib.catseg_stem7_supported(H, W, w.shape[0]))


def stem7_fwd(x, w, bias, bn_stats=False):
    """x: the image (NCHW or NHWC-4, any strides); w: [64, 3, 7, 7] in channels_last memory (physical OHWI).  Returns y NHWC [B, Ho, Wo, 64]
    (+ the BatchNorm partials (part, rows, 0, counts) for bn_finalize)"""
    B, H, W, sb, sc, sy, sx = _image_strides(x)
    Cout = w.shape[0]
    Ho, Wo = conv_out_size(H, 7, 2, 3, 1), conv_out_size(W, 7, 2, 3, 1)
    out = new_act(B, Ho, Wo, Cout, x.device)
    part = cnt = None
    nt = 0
    if bn_stats:
        nt = lib.catseg_stem7_partial_rows(B, H, W)
        part = _bn_part_buffer(3 * nt * Cout + nt, x.device)
        cnt = part[3 * nt * Cout:3 * nt * Cout + nt].view(torch.int32)
    with _Timed("hbm:stem7", 4.0 * (B * 3 * H * W + out.numel())):
        check(lib.catseg_stem7_fwd(ptr(x), sb, sc, sy, sx, B, H, W, ptr(w), ptr(bias), ptr(out), ld_of(out), ptr(part), ptr(cnt), stream()))
    return (out, (part, nt, 0, cnt)) if bn_stats else out


def _refuse_h2_only(t, what):
    """a tensor that exists ONLY as blocked f16x2 planes (concat_bilinear_h2: the HRNet head input) is an unwritten fp32 placeholder too: only
    the f16x2 kernels, through the planes registered for it, may consume it"""
    if getattr(t, "_h2_only", False):
        raise RuntimeError("a tensor that exists only as blocked f16x2 planes reached %s, which reads fp32 (its fp32 tensor was never written)" % what)


def conv_fwd(x, w_ptr_tensor, bias, Cout, kh, kw, stride=1, pad=0, dil=1, out=None, zero_to=0, stem4=False, groups=1, bn_stats=False, train=False,
             exact=False):
    """train=True (the engine's recorded forward): a backward pass will follow -- the split planes of x are written in both layouts
    and kept for it (release_b3_cache() frees them).
    exact=True (the first layers of a trunk, whose rounding error the rest of the network amplifies most: tools/error_growth.py): the
    layer runs the fp32 MFMA kernel (exact operands, two-level accumulation: csrc/igemm.hip TWO_LEVEL) whatever the split-precision
    kernels could take.  (Measured at 2 x 3 x 544 x 960, relative RMS error of layer1's output against fp64: fp32 CPU 1.22e-6, this 0.89e-6,
    two fp16 planes 1.12e-6, three bf16 planes / six products 1.25e-6 -- 84 accumulator roundings per output instead of 42.)
    bn_stats=True: returns (out, partials) where partials = (buffer, n_tiles, tile_rows) are the per-(M-tile, channel)
    BatchNorm partial sums written by the convolution's epilogue (for bn_finalize), or None if this layer's kernel has none"""
    B, H, W, Cin = x.shape
    Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
    if out is None:
        out = new_act(B, Ho, Wo, Cout, x.device, ld=max(zero_to, (Cout + 3) // 4 * 4))
    flops = 2.0 * B * Ho * Wo * Cout * (3 if stem4 else Cin // groups) * kh * kw
    rows = B * Ho * Wo
    part = tr = nt = None
    if bn_stats:
        part = _bn_part_buffer(3 * ((rows + 63) // 64) * Cout, x.device)
        tr, nt = ctypes.c_int(0), ctypes.c_int(0)
    if not exact and not stem4 and zero_to == 0 and w_ptr_tensor.dim() == 4 and _d3_ok(rows, Cin, Cout, kh, kw, stride, pad, dil, groups):
        if bias is None and planes_ok(Cin, rows) and (planes_of(x) is not None or amax_of(x) is not None):
            xp = planes_of(x)
            if xp is None:      # a tensor whose producer wrote no planes (the first block behind a transition): one split pass, kept for backward
                xp = planes_from_f32(x, rec=amax_of(x))
                x._planes = xp
            yrec = new_amax(x.device)
            with _Timed("fwd_d3p", flops):
                res = dconv3_pl(xp, dconv3_weight_image(w_ptr_tensor, h2=True), None, out=out, bn_stats=bn_stats, out_rec=yrec)
            (res[0] if bn_stats else res)._yrec = yrec
            return res
        _refuse_placeholder(x, "the in-kernel-split direct 3x3 kernel")
        rec = amax_of(x) if _trunk_h2() else None
        wimg = dconv3_weight_image(w_ptr_tensor, h2=rec is not None)
        with _Timed("fwd_d3h" if rec is not None else "fwd_d3", flops):
            res = dconv3(x, wimg, bias, out=out, bn_stats=bn_stats, x_amax=rec)
        return res
    _refuse_placeholder(x, "a convolution forward outside the planes route")
    if getattr(x, "_h2_only", False) and not (not exact and "fwd" in B3_OPS and not stem4 and groups == 1 and w_ptr_tensor.dim() == 4 and amax_of(x) is None
                                               and _b3_eligible(rows, Cout, kh * kw, Cin) and _h2()
                                               and _b3_blocked_ok(max(zero_to, Cout), Cin, B * H * W, Cout, kh * kw)):
        _refuse_h2_only(x, "a convolution forward outside the blocked f16x2 route")
    if (not exact and not stem4 and zero_to == 0 and w_ptr_tensor.dim() == 4 and "fwd" in P1_OPS and amax_of(x) is not None
            and _p1_ok(rows, Cin, Cout, kh, kw, stride, pad, dil, groups) and lib.catseg_pconv1_supported(Cout, Cin)
            and rows * ld_of(x) * 4 < B3_PLANE_LIMIT):
        with _Timed("fwd_p1", flops):
            res = pconv1(x, p1_weight_image(w_ptr_tensor), bias, Cout, out, bn_stats=bn_stats)
        drop_amax(out)
        return res
    if (not exact and not stem4 and zero_to == 0 and w_ptr_tensor.dim() == 4 and "fwd" in G1_OPS and amax_of(x) is not None
            and _g1_ok(rows, Cin, Cout, kh, kw, stride, pad, dil, groups) and rows_of(x) * ld_of(x) * 4 < B3_PLANE_LIMIT):
        d = make_desc(x.shape, ld_of(x), Cout, ld_of(out), kh, kw, stride, pad, dil)
        if lib.catseg_gconv_supported(ctypes.byref(d)):
            if bn_stats:
                part = _bn_part_buffer(3 * ((rows + 255) // 256) * Cout, x.device)
            wimg = g1_weight_image(w_ptr_tensor, False, stride, pad, dil)
            with _Timed("fwd_s2p", flops):
                check(lib.catseg_gconv_fwd(ctypes.byref(d), ptr(x), ptr(amax_of(x)), ptr(wimg[0]), ptr(wimg[1]), ptr(bias), ptr(out), ptr(part),
                                           part.numel() if part is not None else 0, ctypes.byref(tr) if bn_stats else None,
                                           ctypes.byref(nt) if bn_stats else None, stream()))
            drop_amax(out)
            if bn_stats:
                return out, ((part, nt.value, tr.value) if tr.value > 0 else None)
            return out
    if not exact and "fwd" in B3_OPS and not stem4 and groups == 1 and w_ptr_tensor.dim() == 4 and _b3_eligible(rows, Cout, kh * kw, Cin):
        d = make_desc(x.shape, Cin, Cout, ld_of(out), kh, kw, stride, pad, dil)
        blk = _b3_blocked_ok(max(zero_to, Cout), Cin, B * H * W, Cout, kh * kw)
        if blk and _h2():
            with _Timed("split3", 0.0):
                xp, xsc = _split3_cached(x, "h2", both=train and not H2T_BLOCKED, keep=train)
                wp, wsc = split2h_weight_blocked(w_ptr_tensor)
            with _Timed("fwd_h2", flops):
                check(lib.catseg_conv2d_fwd_f16x2_blocked(ctypes.byref(d), ptr(xp), ptr(xsc), ptr(wp), ptr(wsc), ptr(bias), ptr(out), zero_to,
                                                          ptr(part), part.numel() if part is not None else 0,
                                                          ctypes.byref(tr) if bn_stats else None, ctypes.byref(nt) if bn_stats else None, stream()))
            if bn_stats:
                return out, ((part, nt.value, tr.value) if tr.value > 0 else None)
            return out
        with _Timed("split3", 0.0):
            xp = _split3_cached(x, "blk" if blk else "planar", both=blk and train, keep=train)
            wp = split3_weight_blocked(w_ptr_tensor) if blk else split3_weight(w_ptr_tensor)
        with _Timed("fwd_b3", flops):
            if blk:
                check(lib.catseg_conv2d_fwd_bf16x3_blocked(ctypes.byref(d), ptr(xp), ptr(wp), ptr(bias), ptr(out), zero_to, ptr(part),
                                                           part.numel() if part is not None else 0,
                                                           ctypes.byref(tr) if bn_stats else None, ctypes.byref(nt) if bn_stats else None, stream()))
            elif bn_stats:
                check(lib.catseg_conv2d_fwd_bf16x3_bnstats(ctypes.byref(d), ptr(xp), ptr(wp), ptr(bias), ptr(out), zero_to, ptr(part),
                                                           part.numel(), ctypes.byref(tr), ctypes.byref(nt), stream()))
            else:
                check(lib.catseg_conv2d_fwd_bf16x3(ctypes.byref(d), ptr(xp), ptr(wp), ptr(bias), ptr(out), zero_to, stream()))
    else:
        d = make_desc(x.shape, ld_of(x), Cout, ld_of(out), kh, kw, stride, pad, dil, stem4, groups)
        with _Timed("fwd", flops):
            if bn_stats:
                check(lib.catseg_conv2d_fwd_bnstats(ctypes.byref(d), ptr(x), ptr(w_ptr_tensor), ptr(bias), ptr(out), zero_to, ptr(part),
                                                    part.numel(), ctypes.byref(tr), ctypes.byref(nt), stream()))
            else:
                check(lib.catseg_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(w_ptr_tensor), ptr(bias), ptr(out), zero_to, stream()))
    if bn_stats:
        return out, ((part, nt.value, tr.value) if tr.value > 0 else None)
    return out


def bn_finalize(partials, rows, C, gamma, eps, momentum, running_mean, running_var, bound=None):
    """batch statistics from the convolution epilogue's per-tile partials: (stats [mean(C), invstd(C)], scale).
    bound = (beta, y_record, z_record): also leaves the bound of the normalised output in z_record (csrc/norm.hip: the exponent of z's planes)"""
    part, n_tiles, tile_rows = partials[:3]
    stats = torch.empty(2 * C, dtype=torch.float32, device=part.device)
    scale = torch.empty(C, dtype=torch.float32, device=part.device)
    if bound is not None:
        beta, yrec, zrec = bound
        check(lib.catseg_bn_finalize_counts_bound(ptr(part), n_tiles, ptr(partials[3]), rows, C, ptr(gamma), ptr(beta), eps, momentum,
                                                  ptr(running_mean), ptr(running_var), ptr(stats), ptr(scale), ptr(yrec), ptr(zrec), stream()))
        return stats, scale
    if len(partials) > 3:       # tiles with individual pixel counts (the direct 3x3 kernel's 2-D tiles)
        check(lib.catseg_bn_finalize_counts(ptr(part), n_tiles, ptr(partials[3]), rows, C, ptr(gamma), eps, momentum, ptr(running_mean),
                                            ptr(running_var), ptr(stats), ptr(scale), stream()))
        return stats, scale
    check(lib.catseg_bn_finalize(ptr(part), n_tiles, tile_rows, rows, C, ptr(gamma), eps, momentum, ptr(running_mean), ptr(running_var),
                                 ptr(stats), ptr(scale), stream()))
    return stats, scale


BN_BWD_FUSE = True    # first pass of the backward of relu(bn1(.)) in the epilogue of the backward-data kernel that produces its dz


def conv_bwd_data(dy, w, xshape, kh, kw, stride=1, pad=0, dil=1, out=None, accumulate=False, groups=1, bn_src=None):
    """bn_src = (q, stats, gamma, beta): the convolution's input was relu(bn(q)) and this call is the ONLY contribution to its
    gradient.  When the direct kernel takes the layer, the result is then already masked (g = dx where relu(bn(q)) > 0) and the
    call returns (g, (partials, n_tiles)) for bn_backward_pre; otherwise it returns dx alone, as without bn_src."""
    B, H, W, Cin = xshape
    Cout = dy.shape[-1]
    _refuse_placeholder(dy, "conv_bwd_data (fp32 output gradient)")
    if out is None:
        out = new_act(B, H, W, Cin, dy.device)
        accumulate = False
    drop_amax(out)
    flops = 2.0 * rows_of(dy) * Cout * (Cin // groups) * kh * kw
    if w.dim() == 4 and _d3_ok(B * H * W, Cin, Cout, kh, kw, stride, pad, dil, groups):
        rec = amax_of(dy) if _trunk_h2() else None
        wimg = dconv3_weight_image(w, backward_data=True, h2=rec is not None)
        kind = "dgrad_d3h" if rec is not None else "dgrad_d3"
        if bn_src is not None and BN_BWD_FUSE and not accumulate:
            q, stats, gamma, beta = bn_src
            nt = lib.catseg_dconv3_tiles(Cin, B, H, W, None, None)
            part = torch.empty(2 * nt * Cin, dtype=torch.float32, device=dy.device)
            with _Timed(kind, flops):
                if rec is not None:
                    check(lib.catseg_dconv3_bnbwd_f16x2(B, H, W, Cin, ptr(dy), ld_of(dy), ptr(rec), ptr(wimg[0]), ptr(wimg[1]), ptr(out), ld_of(out),
                                                        ptr(q), ld_of(q), ptr(stats), ptr(gamma), ptr(beta), ptr(part), part.numel(), stream()))
                else:
                    check(lib.catseg_dconv3_bnbwd(B, H, W, Cin, ptr(dy), ld_of(dy), ptr(wimg), ptr(out), ld_of(out), ptr(q), ld_of(q),
                                                  ptr(stats), ptr(gamma), ptr(beta), ptr(part), part.numel(), stream()))
            return out, (part, nt)
        with _Timed(kind, flops):
            dconv3(dy, wimg, None, out=out, accumulate=accumulate, x_amax=rec)
        return out
    if (w.dim() == 4 and "dgrad" in P1_OPS and amax_of(dy) is not None and _p1_ok(B * H * W, Cout, Cin, kh, kw, stride, pad, dil, groups)
            and lib.catseg_pconv1_supported(Cin, Cout) and rows_of(dy) * ld_of(dy) * 4 < B3_PLANE_LIMIT):
        with _Timed("dgrad_p1", flops):
            pconv1(dy, p1_weight_image(w, transposed=True), None, Cin, out, accumulate=accumulate)
        return out
    if (w.dim() == 4 and "dgrad" in G1_OPS and amax_of(dy) is not None and Cout % 8 == 0 and Cin >= G1_DGRAD_MIN_CIN and _g1_ok(rows_of(dy), Cin, Cout, kh, kw, stride, pad, dil, groups)
            and not _b3_eligible(B * H * W, Cin, kh * kw, (Cout + 7) // 8 * 8, stride == 1) and rows_of(dy) * ld_of(dy) * 4 < B3_PLANE_LIMIT):
        d = make_desc(xshape, ld_of(out), Cout, ld_of(dy), kh, kw, stride, pad, dil)
        dT = _g1_desc(Cout, Cin, kh, kw, stride, pad, dil)      # (the class launches are GEMMs with N = Cin, K = taps * Cout)
        if (lib.catseg_gconv_supported(ctypes.byref(d)) and lib.catseg_pconv1_supported(Cin, ((kh + stride - 1) // stride) * ((kw + stride - 1) // stride) * Cout)
                and d.Ho == dy.shape[1] and d.Wo == dy.shape[2]):
            wimg = g1_weight_image(w, True, stride, pad, dil)
            with _Timed("dgrad_s2p", flops):
                check(lib.catseg_gconv_bwd_data(ctypes.byref(d), ptr(dy), ptr(amax_of(dy)), ptr(wimg[0]), ptr(wimg[1]), ptr(out),
                                                1 if accumulate else 0, stream()))
            del dT
            return out
    if groups == 1 and "dgrad" in B3_OPS and _b3_eligible(B * H * W, Cin, kh * kw, (Cout + 7) // 8 * 8, stride == 1):
        d = make_desc(xshape, ld_of(out), Cout, (Cout + 7) // 8 * 8, kh, kw, stride, pad, dil)
        blk = _b3_blocked_ok(Cin, (Cout + 15) // 16 * 16, rows_of(dy), Cin, kh * kw)
        if blk and _h2():
            with _Timed("split3", 0.0):
                dyp, dysc = _split3_cached_dy(dy, "h2")
                wtp, wtsc = split2h_weight_t_blocked(w)
            with _Timed("dgrad_h2", flops):
                check(lib.catseg_conv2d_bwd_data_f16x2_blocked(ctypes.byref(d), ptr(dyp), ptr(dysc), ptr(wtp), ptr(wtsc), ptr(out),
                                                               1 if accumulate else 0, stream()))
            return out
        with _Timed("split3", 0.0):
            dyp = _split3_cached_dy(dy, "blk" if blk else "planar")
            wtp = split3_weight_t_blocked(w) if blk else split3_weight_t(w)
        with _Timed("dgrad_b3", flops):
            fn = lib.catseg_conv2d_bwd_data_bf16x3_blocked if blk else lib.catseg_conv2d_bwd_data_bf16x3
            check(fn(ctypes.byref(d), ptr(dyp), ptr(wtp), ptr(out), 1 if accumulate else 0, stream()))
        return out
    d = make_desc(xshape, ld_of(out), Cout, ld_of(dy), kh, kw, stride, pad, dil, False, groups)
    with _Timed("dgrad", flops):
        check(lib.catseg_conv2d_bwd_data(ctypes.byref(d), ptr(dy), ptr(w), ptr(out), 1 if accumulate else 0, stream()))
    return out


def _wgrad_split_route(x, dy, kh, kw, stride, stem4, groups):
    """backward-weight of this layer runs the split-precision implicit GEMM (igemm_h2t / igemm_b3t) on planes of x and dy"""
    Cout, Cin = dy.shape[-1], x.shape[-1]
    return bool(groups == 1 and "wgrad" in B3_OPS and not stem4 and PRECISION == "bf16x3" and Cin % 8 == 0 and
                rows_of(x) * Cin < B3_INDEX_LIMIT and rows_of(dy) * ((Cout + 7) // 8 * 8) < B3_INDEX_LIMIT and
                ((B3_MIN_TAPS <= kh * kw and kh * kw * Cin >= B3_MIN_K and Cout >= B3_MIN_N and rows_of(dy) >= B3_MIN_WGRAD_ROWS)
                 or (stride == 1 and _b3_wide_1x1(rows_of(dy), Cout, kh * kw, Cin))))


def _wgrad_dgrad_blk(x, dy, kh, kw, stride):
    """the layer's backward-data will read BLOCKED planes of dy: the backward-weight's split pass writes them as well"""
    Cout, Cin = dy.shape[-1], x.shape[-1]
    return bool(stride == 1 and "dgrad" in B3_OPS and _b3_eligible(rows_of(x), Cin, kh * kw, (Cout + 7) // 8 * 8)
                and _b3_blocked_ok(Cin, (Cout + 15) // 16 * 16, rows_of(dy), Cin, kh * kw))


def _wgrad_h2_route(x, dy):
    Cout, Cin = dy.shape[-1], x.shape[-1]
    pad = 16 if H2T_BLOCKED else 8
    return bool(_h2() and 4 * rows_of(x) * ((Cin + pad - 1) // pad * pad) < B3_PLANE_LIMIT
                and 4 * rows_of(dy) * ((Cout + pad - 1) // pad * pad) < B3_PLANE_LIMIT)


def _wgrad_h2_planes(x, dy, dgrad_blk):
    """(x planes, x scale, dy planes, dy scale) of the f16x2 backward-weight kernel"""
    if H2T_BLOCKED:
        xp, xsc = _split3_cached(x, "h2")
        dyp, dysc = _split3_cached_dy(dy, "h2")
    else:
        xp, xsc = _split3_cached(x, "h2p")
        dyp, dysc = _split3_cached_dy(dy, "h2p", both=dgrad_blk)
    return xp, xsc, dyp, dysc


def conv_bwd_weight(x, dy, dw, dbias, kh, kw, stride=1, pad=0, dil=1, stem4=False, groups=1):
    """dw: destination tensor (physical OHWI, or packed [O][7][8][4] for the stem)."""
    Cout, Cin = dy.shape[-1], x.shape[-1]
    _refuse_placeholder(x, "conv_bwd_weight (fp32 input)")
    _refuse_placeholder(dy, "conv_bwd_weight (fp32 output gradient)")
    if getattr(x, "_h2_only", False) and not (H2T_BLOCKED and not stem4 and amax_of(x) is None and _wgrad_split_route(x, dy, kh, kw, stride, stem4, groups)
                                               and _wgrad_h2_route(x, dy)
                                               and not (_d3_ok(rows_of(dy), Cin, Cout, kh, kw, stride, pad, dil, groups) and lib.catseg_dwgrad3_supported(Cin))):
        _refuse_h2_only(x, "conv_bwd_weight outside the blocked f16x2 route")
    flops = 2.0 * rows_of(dy) * Cout * (3 if stem4 else Cin // groups) * kh * kw
    if (not stem4 and x.dim() == 4 and _d3_ok(rows_of(dy), Cin, Cout, kh, kw, stride, pad, dil, groups)
            and lib.catseg_dwgrad3_supported(Cin)):
        dwgrad3(x, dy, dw, dbias, flops)
        return dw
    if (not stem4 and x.dim() == 4 and dw.dim() == 4 and "wgrad" in P1_OPS and amax_of(x) is not None and amax_of(dy) is not None
            and min(Cout, Cin) >= P1_WGRAD_MIN_DIM
            and _p1_ok(rows_of(dy), Cin, Cout, kh, kw, stride, pad, dil, groups) and lib.catseg_pconv1_wgrad_supported(Cout, Cin)
            and rows_of(x) * ld_of(x) * 4 < B3_PLANE_LIMIT and rows_of(dy) * ld_of(dy) * 4 < B3_PLANE_LIMIT):
        need = lib.catseg_pconv1_wgrad_workspace(rows_of(dy), Cout, Cin)
        ws = workspace(need + 256 * Cout * 4, x.device)
        with _Timed("wgrad_p1", flops):
            check(lib.catseg_pconv1_wgrad(rows_of(dy), Cout, Cin, ptr(dy), ld_of(dy), ptr(amax_of(dy)), ptr(x), ld_of(x), ptr(amax_of(x)), ptr(dw),
                                          ptr(ws), need, stream()))
        if dbias is not None:
            check(lib.catseg_bias_grad(ptr(dy), ld_of(dy), rows_of(dy), Cout, ptr(dbias), ptr(ws), ws.numel(), stream()))
        return dw
    if (not stem4 and x.dim() == 4 and dw.dim() == 4 and "wgrad" in G1_OPS and amax_of(x) is not None and amax_of(dy) is not None
            and _g1_ok(rows_of(dy), Cin, Cout, kh, kw, stride, pad, dil, groups) and not _wgrad_split_route(x, dy, kh, kw, stride, stem4, groups)
            and rows_of(x) * ld_of(x) * 4 < B3_PLANE_LIMIT and rows_of(dy) * ld_of(dy) * 4 < B3_PLANE_LIMIT):
        d = make_desc(x.shape, ld_of(x), Cout, ld_of(dy), kh, kw, stride, pad, dil)
        if lib.catseg_gconv_wgrad_supported(ctypes.byref(d)) and d.Ho == dy.shape[1] and d.Wo == dy.shape[2]:
            need = lib.catseg_gconv_wgrad_workspace(ctypes.byref(d))
            ws = workspace(need + 256 * Cout * 4, x.device)
            with _Timed("wgrad_s2p", flops):
                check(lib.catseg_gconv_bwd_weight(ctypes.byref(d), ptr(dy), ptr(amax_of(dy)), ptr(x), ptr(amax_of(x)), ptr(dw), ptr(ws), need, stream()))
            if dbias is not None:
                check(lib.catseg_bias_grad(ptr(dy), ld_of(dy), rows_of(dy), Cout, ptr(dbias), ptr(ws), ws.numel(), stream()))
            return dw
    if _wgrad_split_route(x, dy, kh, kw, stride, stem4, groups):
        d = make_desc(x.shape, Cin, Cout, (Cout + 7) // 8 * 8, kh, kw, stride, pad, dil)
        ws = workspace(lib.catseg_conv2d_bwd_weight_bf16x3_workspace(ctypes.byref(d)) + 256 * Cout * 4, x.device)
        dgrad_blk = _wgrad_dgrad_blk(x, dy, kh, kw, stride)
        if _wgrad_h2_route(x, dy):
            # two fp16 planes per operand (csrc/igemm_f16x2.hip; each operand's planes behind one 32-bit-offset buffer resource); dy's
            # blocked planes for this layer's backward-data from the same pass
            wsb = workspace(lib.catseg_conv2d_bwd_weight_f16x2_workspace(ctypes.byref(d)) + 256 * Cout * 4, x.device)
            with _Timed("split3", 0.0):
                xp, xsc, dyp, dysc = _wgrad_h2_planes(x, dy, dgrad_blk)
            with _Timed("wgrad_h2", flops):
                fn = lib.catseg_conv2d_bwd_weight_f16x2_blocked if H2T_BLOCKED else lib.catseg_conv2d_bwd_weight_f16x2
                check(fn(ctypes.byref(d), ptr(xp), ptr(xsc), ptr(dyp), ptr(dysc), ptr(dw), ptr(wsb), wsb.numel(), stream()))
            if dbias is not None:
                check(lib.catseg_bias_grad(ptr(dy), ld_of(dy), rows_of(dy), Cout, ptr(dbias), ptr(wsb), wsb.numel(), stream()))
            return dw
        with _Timed("split3", 0.0):
            xp = _split3_cached(x, "planar")
            dyp = _split3_cached_dy(dy, "planar", both=dgrad_blk)
        with _Timed("wgrad_b3", flops):
            check(lib.catseg_conv2d_bwd_weight_bf16x3(ctypes.byref(d), ptr(xp), ptr(dyp), ptr(dw), ptr(ws), ws.numel(), stream()))
        if dbias is not None:
            check(lib.catseg_bias_grad(ptr(dy), ld_of(dy), rows_of(dy), Cout, ptr(dbias), ptr(ws), ws.numel(), stream()))
        return dw
    d = make_desc(x.shape, ld_of(x), Cout, ld_of(dy), kh, kw, stride, pad, dil, stem4, groups)
    need = lib.catseg_conv2d_bwd_weight_workspace(ctypes.byref(d))
    ws = workspace(need, x.device)
    with _Timed("wgrad", flops):
        check(lib.catseg_conv2d_bwd_weight(ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(dbias), ptr(ws), ws.numel(), stream()))
    return dw


def dwgrad3(x, dy, dw, dbias=None, flops=0.0):
    """backward-weight of a 3x3 / stride 1 / pad 1 trunk convolution through the direct split-precision kernel"""
    B, H, W, C = x.shape
    need = lib.catseg_dwgrad3_workspace(B, H, W, C)
    ws = workspace(need + 256 * C * 4, x.device)
    rx, rd = (amax_of(x), amax_of(dy)) if _trunk_h2() else (None, None)
    if rx is not None and rd is not None:      # both operands carry their producers' amax records: two fp16 planes, three products
        with _Timed("wgrad_d3h", flops or 2.0 * B * H * W * C * C * 9):
            check(lib.catseg_dwgrad3_f16x2(B, H, W, C, ptr(x), ld_of(x), ptr(rx), ptr(dy), ld_of(dy), ptr(rd), ptr(dw), ptr(ws), need, stream()))
    else:
        with _Timed("wgrad_d3", flops or 2.0 * B * H * W * C * C * 9):
            check(lib.catseg_dwgrad3(B, H, W, C, ptr(x), ld_of(x), ptr(dy), ld_of(dy), ptr(dw), ptr(ws), need, stream()))
    if dbias is not None:
        check(lib.catseg_bias_grad(ptr(dy), ld_of(dy), rows_of(dy), C, ptr(dbias), ptr(ws), ws.numel(), stream()))
    return dw


NT, NN, TN = 0, 1, 2


def gemm(layout, batch, M, N, K, A, lda, sA, Bm, ldb, sB, Cm, ldc, sC, zero_to=0, accumulate=False):
    check(lib.catseg_gemm_batched(layout, batch, M, N, K, ptr(A), lda, sA, ptr(Bm), ldb, sB, ptr(Cm), ldc, sC, zero_to,
                                  1 if accumulate else 0, stream()))
    return Cm


GEMM_TN_SPLIT = _plan.get("gemm_tn_split")


def tn_splits(M, N, K):
    """row chunks of a long reduction into a small M x N result: the largest divisor of K up to 32 that leaves chunks of >= 512 rows (a
    multiple of 4: the chunk's byte offset stays 16-byte aligned); 1 = do not split (short K, a large result, no suitable divisor)"""
    if K >= 4096 and M * N <= 64 * 1024:
        for s in (32, 24, 20, 16, 15, 12, 10, 8, 6, 5, 4, 3, 2):
            if K % s == 0 and K // s >= 512 and (K // s) % 4 == 0:
                return s
    return 1


def gemm_tn_split(batch, M, N, K, A, lda, Bm, ldb, Cm, accumulate=False):
    """C[b] (+)= A[b]^T B[b] for a LONG reduction K into a SMALL result M x N (the OCR head: all H * W pixels of an image into K_classes x C;
    models/OCR.py:158-170, and the value / key gradients of :266-274): with one block chain per (image, column tile) the 32 640-row
    reduction of the bench shape runs on 64 of the chip's 256 CUs for ~2 ms; the rows are cut into `splits` chunks that run as batch * splits
    independent GEMMs and are summed in a fixed order (catseg_sum_slabs).  A, Bm: [batch, K, ld] contiguous; Cm: [batch, M, N] contiguous."""
    splits = tn_splits(M, N, K) if GEMM_TN_SPLIT else 1
    if splits == 1:
        return gemm(TN, batch, M, N, K, A, lda, K * lda, Bm, ldb, K * ldb, Cm, N, M * N, accumulate=accumulate)
    kc = K // splits
    part = torch.empty((batch * splits, M, N), dtype=torch.float32, device=Cm.device)
    gemm(TN, batch * splits, M, N, kc, A, lda, kc * lda, Bm, ldb, kc * ldb, part, N, M * N)
    check(lib.catseg_sum_slabs(ptr(part), ptr(Cm), M * N, splits, batch, 1 if accumulate else 0, stream()))
    return Cm


def bn_train_stats(y, gamma, eps, momentum, running_mean, running_var):
    C = y.shape[-1]
    rows = rows_of(y)
    stats = torch.empty(2 * C, dtype=torch.float32, device=y.device)
    scale = torch.empty(C, dtype=torch.float32, device=y.device)
    ws = workspace(lib.catseg_bn_workspace(rows, C), y.device)
    check(lib.catseg_bn_train_stats(ptr(y), rows, C, ld_of(y), ptr(gamma), eps, momentum, ptr(running_mean),
                                    ptr(running_var), ptr(stats), ptr(scale), ptr(ws), ws.numel(), stream()))
    return stats, scale


def bn_eval_scale(gamma, running_var, eps):
    scale = torch.empty_like(gamma)
    check(lib.catseg_bn_eval_scale(gamma.numel(), ptr(gamma), ptr(running_var), eps, ptr(scale), stream()))
    return scale


# The ReLU mask of a residual block's output as bits (csrc/norm.hip: catseg_bn_apply_mask / catseg_bn_backward_mask): the BatchNorm backward of
# z = relu(bn(y) + residual) reads 1 bit per element instead of z in both of its passes.  CATSEG_RELU_BITS=0: z is read, as before.
RELU_BITS = _plan.get("relu_bits")


def relu_bits_ok(y, residual, relu, out):
    return bool(RELU_BITS and relu and residual is not None and y.is_cuda and y.shape[-1] % 8 == 0 and ld_of(y) % 4 == 0
                and (out is None or ld_of(out) % 4 == 0))


def bn_apply(y, mean, scale, beta, residual, relu, out=None, planes_rec=None, planes_only=False, want_mask=False):
    """the output carries an amax record (out._amax: max|out| accumulated by the kernel) when the trunk runs the f16x2 kernels.
    planes_rec (the record bn_finalize(bound=...) left the bound in): the kernel also writes the fp16 x 2 planes of the output (out._planes);
    planes_only: and NOT the fp32 output -- `out` is then an unwritten placeholder that only plane-streaming kernels may consume"""
    if out is None:
        out = torch.empty(y.shape, dtype=torch.float32, device=y.device)
    if planes_rec is not None:
        buf = torch.empty(lib.catseg_planes_bytes(rows_of(y), y.shape[-1]), dtype=torch.uint8, device=y.device)
        mask = None
        if want_mask and not planes_only and relu_bits_ok(y, residual, relu, out):
            mask = torch.empty(lib.catseg_bn_mask_bytes(rows_of(y), y.shape[-1]), dtype=torch.uint8, device=y.device)
        # algorithmic bytes: y (+ residual) read, planes written (4 B / element, like fp32), z written unless planes only
        with _Timed("hbm:bn_apply", 4.0 * y.numel() * ((3 if residual is not None else 2) + (0 if planes_only else 1))):
            if mask is not None:
                check(lib.catseg_bn_apply_planes_mask(ptr(y), ld_of(y), ptr(mean), ptr(scale), ptr(beta), ptr(residual), ld_of(residual),
                                                      ptr(amax_of(residual)), ptr(out), ld_of(out), ptr(buf), rows_of(y), y.shape[-1], ptr(planes_rec),
                                                      ptr(mask), stream()))
                out._relu_mask = mask
            else:
                check(lib.catseg_bn_apply_planes(ptr(y), ld_of(y), ptr(mean), ptr(scale), ptr(beta), ptr(residual),
                                                 ld_of(residual) if residual is not None else 0, ptr(amax_of(residual)) if residual is not None else None,
                                                 None if planes_only else ptr(out), ld_of(out), ptr(buf), rows_of(y), y.shape[-1], 1 if relu else 0,
                                                 ptr(planes_rec), stream()))
        out._amax = planes_rec
        out._planes = Planes(buf, planes_rec, y.shape)
        out._planes_only = bool(planes_only)
        return out
    if want_mask and relu_bits_ok(y, residual, relu, out):
        rows, C = rows_of(y), y.shape[-1]
        mask = torch.empty(lib.catseg_bn_mask_bytes(rows, C), dtype=torch.uint8, device=y.device)
        rec = new_amax(y.device) if _trunk_h2() else None
        with _Timed("hbm:bn_apply", 4.0 * y.numel() * 3):
            check(lib.catseg_bn_apply_mask(ptr(y), ld_of(y), ptr(mean), ptr(scale), ptr(beta), ptr(residual), ld_of(residual), ptr(out), ld_of(out),
                                           rows, C, ptr(rec), ptr(mask), stream()))
        if rec is not None:
            out._amax = rec
        out._relu_mask = mask
        return out
    with _Timed("hbm:bn_apply", 4.0 * y.numel() * (3 if residual is not None else 2)):
        _bn_apply(y, mean, scale, beta, residual, relu, out)
    return out


def bn_backward_planes(dz, z, y, stats, gamma, relu, dgamma, dbeta, dres, dres_accumulate, beta, y_rec):
    """bn_backward whose result exists as planes only (returned: Planes); y_rec: the forward convolution's max|y| record"""
    C = y.shape[-1]
    rows = rows_of(y)
    ws = workspace(lib.catseg_bn_workspace(rows, C), y.device)
    buf = torch.empty(lib.catseg_planes_bytes(rows, C), dtype=torch.uint8, device=y.device)
    rec, grec = new_amax(y.device), new_amax(y.device)
    mask = getattr(z, "_relu_mask", None) if (z is not None and relu) else None
    if mask is not None:
        with _Timed("hbm:bn_backward", 4.0 * y.numel() * (5 + (1 if dres is not None else 0))):
            check(lib.catseg_bn_backward_planes_mask(ptr(dz), ld_of(dz), ptr(mask), ptr(y), ld_of(y), ptr(stats), ptr(gamma), rows, C, ptr(buf), ptr(rec),
                                                     ptr(grec), ptr(y_rec), ptr(dgamma), ptr(dbeta), ptr(dres), ld_of(dres) if dres is not None else 0,
                                                     1 if dres_accumulate else 0, ptr(ws), ws.numel(), stream()))
        return Planes(buf, rec, y.shape)
    with _Timed("hbm:bn_backward", 4.0 * y.numel() * (5 + (1 if dres is not None else 0))):
        check(lib.catseg_bn_backward_planes(ptr(dz), ld_of(dz), ptr(z), ld_of(z) if z is not None else 0, ptr(y), ld_of(y), ptr(stats), ptr(gamma),
                                            ptr(beta), rows, C, 1 if relu else 0, ptr(buf), ptr(rec), ptr(grec), ptr(y_rec), ptr(dgamma), ptr(dbeta),
                                            ptr(dres), ld_of(dres) if dres is not None else 0, 1 if dres_accumulate else 0, ptr(ws), ws.numel(),
                                            stream()))
    return Planes(buf, rec, y.shape)


def bn_backward_pre_planes(g, q, stats, gamma, pre, y_rec, dgamma, dbeta):
    """bn_backward_pre (g already masked, its per-wave sums and max|g| from catseg_dconv3_pl_bnbwd: pre = (partials, rows, g_record)) whose
    result exists as planes only"""
    C = q.shape[-1]
    rows = rows_of(q)
    part, nr, grec = pre
    ws = workspace(lib.catseg_bn_workspace(rows, C), q.device)
    buf = torch.empty(lib.catseg_planes_bytes(rows, C), dtype=torch.uint8, device=q.device)
    rec = new_amax(q.device)
    with _Timed("hbm:bn_backward", 4.0 * q.numel() * 3):
        check(lib.catseg_bn_backward_pre_planes(ptr(g), ld_of(g), ptr(q), ld_of(q), ptr(stats), ptr(gamma), ptr(part), nr, rows, C, ptr(buf), ptr(rec),
                                                ptr(grec), ptr(y_rec), ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel(), stream()))
    return Planes(buf, rec, q.shape)


def conv_bwd_data_pl(dyp, w, out, accumulate=False, bn_src=None):
    """backward-data of a trunk 3x3 convolution from the planes of dy; bn_src as in conv_bwd_data: then returns
    (g, (partials, rows, g_record)) for bn_backward_pre(_planes)"""
    B, H, W, C = dyp.shape
    wimg = dconv3_weight_image(w, backward_data=True, h2=True)
    flops = 2.0 * B * H * W * C * C * 9
    drop_amax(out)
    if bn_src is not None and BN_BWD_FUSE and not accumulate:
        q, stats, gamma, beta = bn_src
        nr = lib.catseg_dconv3_pl_rows(C, B, H, W)
        part = torch.empty(2 * nr * C, dtype=torch.float32, device=out.device)
        grec = new_amax(out.device)
        with _Timed("dgrad_d3p", flops):
            check(lib.catseg_dconv3_pl_bnbwd(B, H, W, C, ptr(dyp.buf), ptr(dyp.rec), ptr(wimg[0]), ptr(wimg[1]), ptr(out), ld_of(out), ptr(q), ld_of(q),
                                             ptr(stats), ptr(gamma), ptr(beta), ptr(part), part.numel(), ptr(grec), stream()))
        return out, (part, nr, grec)
    with _Timed("dgrad_d3p", flops):
        dconv3_pl(dyp, wimg, None, out=out, accumulate=accumulate)
    return out


def _bn_apply(y, mean, scale, beta, residual, relu, out):
    rec = new_amax(y.device) if _trunk_h2() else None
    check(lib.catseg_bn_apply_amax(ptr(y), ld_of(y), ptr(mean), ptr(scale), ptr(beta), ptr(residual),
                                   ld_of(residual) if residual is not None else 0, ptr(out), ld_of(out), rows_of(y),
                                   y.shape[-1], 1 if relu else 0, ptr(rec), stream()))
    if rec is not None:
        out._amax = rec
    return out


def bn_backward(dz, z, y, stats, gamma, relu, dgamma, dbeta, dres=None, dres_accumulate=False, dy_out=None, beta=None):
    """z=None (only without a residual branch): the ReLU mask is recomputed from y and beta instead of being read; a z that carries its mask as
    bits (bn_apply(want_mask=True)) is not read either"""
    C = y.shape[-1]
    rows = rows_of(y)
    if dy_out is None:
        dy_out = torch.empty(y.shape, dtype=torch.float32, device=y.device)
    ws = workspace(lib.catseg_bn_workspace(rows, C), y.device)
    mask = getattr(z, "_relu_mask", None) if (z is not None and relu) else None
    if mask is not None:
        rec = new_amax(y.device) if _trunk_h2() else None
        # algorithmic bytes: two passes over (dz, y) + dy written (+ the residual gradient)
        with _Timed("hbm:bn_backward", 4.0 * y.numel() * (5 + (1 if dres is not None else 0))):
            check(lib.catseg_bn_backward_mask(ptr(dz), ld_of(dz), ptr(mask), ptr(y), ld_of(y), ptr(stats), ptr(gamma), rows, C, ptr(dy_out),
                                              ld_of(dy_out), ptr(dgamma), ptr(dbeta), ptr(dres), ld_of(dres) if dres is not None else 0,
                                              1 if dres_accumulate else 0, ptr(ws), ws.numel(), ptr(rec), stream()))
        if rec is not None:
            dy_out._amax = rec
        return dy_out
    # algorithmic bytes: two passes over (dz, y [or z]) + dy written (+ the residual gradient)
    with _Timed("hbm:bn_backward", 4.0 * y.numel() * (5 + (1 if dres is not None else 0))):
        _bn_backward(dz, z, y, stats, gamma, relu, dgamma, dbeta, dres, dres_accumulate, dy_out, beta, rows, C, ws)
    return dy_out


# The HRNet head input -- torch.cat of branch 0 and the bilinearly upsampled branches 1..3 (models/HRNetv2.py:505-508) -- feeds nothing but the two
# 3 x 3 head convolutions on the f16x2 kernels: ONE launch interpolates, splits and writes the blocked planes (csrc/igemm_f16x2.hip:
# concat_bilinear_split2h_kernel); the fp32 concatenation, the copy of branch 0 into it and the split pass over it are gone.
# CATSEG_CONCAT_PLANES=0: fp32 concatenation + catseg_split2h, as before.
CONCAT_PLANES = _plan.get("concat_planes")


def concat_planes_route(ys, consumers):
    """True when every consumer (the Conv2d modules that read the concatenation of `ys` at ys[0]'s size) runs forward AND backward-weight on
    blocked f16x2 planes of its input: the concatenation then never needs to exist in fp32"""
    import types
    if not (CONCAT_PLANES and H2T_BLOCKED and _h2() and PRECISION == "bf16x3" and consumers and 1 <= len(ys) <= 4):
        return False
    if not all(y.is_cuda and y.dim() == 4 and y.shape[-1] % 16 == 0 and ld_of(y) % 4 == 0 and amax_of(y) is not None and y.shape[0] == ys[0].shape[0]
               for y in ys):
        return False
    B, H, W, _ = ys[0].shape
    Cin, rows = sum(y.shape[-1] for y in ys), B * H * W
    if 4 * rows * Cin >= B3_PLANE_LIMIT:
        return False
    x = types.SimpleNamespace(shape=(B, H, W, Cin))
    for conv in consumers:
        kh, kw = conv.kernel_size
        st, pd, dl, Cout = conv.stride[0], conv.padding[0], conv.dilation[0], conv.out_channels
        if (conv.in_channels != Cin or getattr(conv, "exact_operands", False) or getattr(conv, "stem", False) or conv.groups != 1
                or conv_out_size(H, kh, st, pd, dl) != H or conv_out_size(W, kw, st, pd, dl) != W):
            return False
        if _d3_ok(rows, Cin, Cout, kh, kw, st, pd, dl, 1):
            return False
        if not ("fwd" in B3_OPS and _b3_eligible(rows, Cout, kh * kw, Cin) and _b3_blocked_ok(Cout, Cin, rows, Cout, kh * kw)):
            return False
        dy = types.SimpleNamespace(shape=(B, H, W, Cout))
        if not (_wgrad_split_route(x, dy, kh, kw, st, False, 1) and _wgrad_h2_route(x, dy)):
            return False
    return True


def concat_bilinear_h2(ys, Ho, Wo):
    """(blocked planes [2, sum C / 16, B Ho Wo, 16], scale record) of cat_i(bilinear(y_i -> Ho x Wo, align_corners=False))"""
    n, B, dev = len(ys), ys[0].shape[0], ys[0].device
    C, rows = sum(y.shape[-1] for y in ys), B * Ho * Wo
    blk = torch.empty((2, C // 16, rows, 16), dtype=torch.int16, device=dev)
    scale = torch.empty(2, dtype=torch.int32, device=dev)
    PA, IA = ctypes.c_void_p * 4, ctypes.c_int * 4
    pad = lambda v: list(v) + [0] * (4 - n)
    xs, recs = PA(*pad([ptr(y) for y in ys])), PA(*pad([ptr(amax_of(y)) for y in ys]))
    lds, Hs, Ws, Cs = IA(*pad([ld_of(y) for y in ys])), IA(*pad([y.shape[1] for y in ys])), IA(*pad([y.shape[2] for y in ys])), IA(*pad([y.shape[-1] for y in ys]))
    with _Timed("hbm:bilinear_fwd", 4.0 * (sum(y.numel() for y in ys) + rows * C)):
        check(lib.catseg_concat_bilinear_split2h(n, xs, lds, Hs, Ws, Cs, recs, B, Ho, Wo, ptr(blk), ptr(scale), stream()))
    return blk, scale


def register_h2_planes(x, blk, scale):
    """x: an UNWRITTEN fp32 placeholder whose contents exist as the blocked planes (blk, scale).  The planes travel with the tensor (they live
    as long as the tape holds it: a second forward pass before this one's backward does not release them); _split3_cached hands them to the
    f16x2 kernels where a split pass of x would have run; every fp32 route refuses x"""
    x._h2_only = True
    x._h2_planes = (blk, scale)


# The head layers (conv -> BatchNorm -> ReLU on the f16x2 kernels: models/OCR.py:72-89, 326-333): the BatchNorm backward writes dy straight
# as the blocked fp16 x 2 planes that the layer's backward-weight AND backward-data read (csrc/norm.hip: bn_bwd_apply_h2_kernel) -- no fp32 dy,
# no split pass over it, the bias gradient from the same pass.  CATSEG_HEAD_DY_PLANES=0: fp32 dy + catseg_split2h, as before.
HEAD_DY_PLANES = _plan.get("head_dy_planes")


def h2_dy_route(x, y, w, kh, kw, stride, pad, dil, groups, need_dx):
    """True when conv_bwd_weight / conv_bwd_data would BOTH run this layer on the blocked f16x2 planes of dy (the conditions below repeat
    their route selection in its order; tests/test_heads_dy_planes_gpu.py pins the two against each other)"""
    if not (HEAD_DY_PLANES and H2T_BLOCKED and _h2() and x.is_cuda and x.dim() == 4 and y.dim() == 4 and w.dim() == 4 and groups == 1):
        return False
    Cout, Cin, taps = y.shape[-1], x.shape[-1], kh * kw
    rows_o, rows_i = rows_of(y), rows_of(x)
    if Cout % 64 != 0 or 4 * rows_o * Cout >= B3_PLANE_LIMIT:
        return False
    # backward-weight: direct trunk kernel, pointwise kernel (the gather kernel is excluded by _wgrad_split_route), then the split route
    if _d3_ok(rows_o, Cin, Cout, kh, kw, stride, pad, dil, groups) and lib.catseg_dwgrad3_supported(Cin):
        return False
    if ("wgrad" in P1_OPS and min(Cout, Cin) >= P1_WGRAD_MIN_DIM and _p1_ok(rows_o, Cin, Cout, kh, kw, stride, pad, dil, groups)
            and lib.catseg_pconv1_wgrad_supported(Cout, Cin)):
        return False
    if not (_wgrad_split_route(x, y, kh, kw, stride, False, groups) and _wgrad_h2_route(x, y)):
        return False
    if need_dx:
        if _d3_ok(rows_i, Cin, Cout, kh, kw, stride, pad, dil, groups):
            return False
        if "dgrad" in P1_OPS and _p1_ok(rows_i, Cout, Cin, kh, kw, stride, pad, dil, groups) and lib.catseg_pconv1_supported(Cin, Cout):
            return False
        if not (stride == 1 and "dgrad" in B3_OPS and _b3_eligible(rows_i, Cin, taps, (Cout + 7) // 8 * 8, True)
                and _b3_blocked_ok(Cin, (Cout + 15) // 16 * 16, rows_o, Cin, taps)):
            return False
    return True


def bn_backward_h2(dz, y, stats, gamma, relu, dgamma, dbeta, beta, dbias=None):
    """BatchNorm (+ ReLU, mask recomputed from y) backward without a residual branch: (blocked planes of dy, their scale record); dbias = the
    column sums of dy"""
    C, rows = y.shape[-1], rows_of(y)
    blk = torch.empty((2, C // 16, rows, 16), dtype=torch.int16, device=y.device)
    scale = torch.empty(2, dtype=torch.int32, device=y.device)
    need = lib.catseg_bn_backward_h2_workspace(rows, C)
    ws = workspace(need, y.device)
    grec, yrec, dyrec = new_amax(y.device), new_amax(y.device), new_amax(y.device)
    with _Timed("hbm:bn_backward", 4.0 * y.numel() * 5):
        check(lib.catseg_bn_backward_h2(ptr(dz), ld_of(dz), None, 0, ptr(y), ld_of(y), ptr(stats), ptr(gamma), ptr(beta), rows, C, 1 if relu else 0,
                                        ptr(blk), ptr(scale), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(grec), ptr(yrec), ptr(dyrec), ptr(ws),
                                        ws.numel(), stream()))
    return blk, scale


HEAD_FUSE = _plan.get("head_fuse")


def head_fuse_ok(y, K):
    """the classifier behind a head's BatchNorm + ReLU runs fused with it (csrc/headfuse.h): C % 64 == 0 up to 512 channels, at most 32 classes"""
    C = y.shape[-1]
    return bool(HEAD_FUSE and y.is_cuda and y.dim() == 4 and C % 64 == 0 and C <= 512 and 1 <= K <= 32 and ld_of(y) % 4 == 0
                and 4 * rows_of(y) * C < B3_PLANE_LIMIT)


def head_fwd(y, mean, scale, beta, wh, bh, K, ld):
    """logits [B, H, W, K] (pixel stride ld, zero padded) = relu(bn(y)) wh^T + bh; the normalised activation is not written"""
    B, H, W, C = y.shape
    buf = torch.empty((B, H, W, ld), dtype=torch.float32, device=y.device)
    rows = rows_of(y)
    with _Timed("hbm:head_fwd", 4.0 * y.numel()):
        check(lib.catseg_head_fwd(ptr(y), ld_of(y), ptr(mean), ptr(scale), ptr(beta), ptr(wh), ptr(bh), K, rows, C, ptr(buf), ld, min(ld, 32),
                                  stream()))
    return buf[..., :K] if ld != K else buf


def head_backward(dl, y, stats, gamma, beta, wh, dwh, dbh, dgamma, dbeta, dbias=None):
    """backward of head_fwd: (blocked planes of dy, their scale record) as bn_backward_h2 returns them; dwh / dbh / dgamma / dbeta / dbias written"""
    C, rows, K = y.shape[-1], rows_of(y), wh.shape[0]
    if ld_of(dl) < 32 or ld_of(dl) % 4:        # (a caller's dense gradient: the kernels read 128-byte rows)
        padded = new_act(dl.shape[0], dl.shape[1], dl.shape[2], K, dl.device, ld=32, zero=True)
        padded.copy_(dl)
        dl = padded
    blk = torch.empty((2, C // 16, rows, 16), dtype=torch.int16, device=y.device)
    scale = torch.empty(2, dtype=torch.int32, device=y.device)
    ws = workspace(lib.catseg_head_backward_workspace(rows, C), y.device)
    grec, yrec, dyrec = new_amax(y.device), new_amax(y.device), new_amax(y.device)
    with _Timed("hbm:head_backward", 4.0 * y.numel() * 3):
        check(lib.catseg_head_backward(ptr(dl), ld_of(dl), ptr(y), ld_of(y), ptr(stats), ptr(gamma), ptr(beta), ptr(wh), K, rows, C, ptr(blk),
                                       ptr(scale), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(dwh), ptr(dbh), ptr(grec), ptr(yrec), ptr(dyrec),
                                       ptr(ws), ws.numel(), stream()))
    return blk, scale


def conv_bwd_weight_h2(x, dyp, dysc, Cout, dw, kh, kw, stride, pad, dil):
    """backward-weight from the blocked planes of dy (bn_backward_h2) and of x (kept from the forward pass, or split now)"""
    Cin = x.shape[-1]
    d = make_desc(x.shape, Cin, Cout, (Cout + 7) // 8 * 8, kh, kw, stride, pad, dil)
    wsb = workspace(lib.catseg_conv2d_bwd_weight_f16x2_workspace(ctypes.byref(d)) + 256 * Cout * 4, x.device)
    with _Timed("split3", 0.0):
        xp, xsc = _split3_cached(x, "h2")
    with _Timed("wgrad_h2", 2.0 * d.B * d.Ho * d.Wo * Cout * Cin * kh * kw):
        check(lib.catseg_conv2d_bwd_weight_f16x2_blocked(ctypes.byref(d), ptr(xp), ptr(xsc), ptr(dyp), ptr(dysc), ptr(dw), ptr(wsb), wsb.numel(),
                                                         stream()))
    return dw


def conv_bwd_data_h2(dyp, dysc, w, xshape, Cout, kh, kw, pad, dil, out, accumulate):
    """backward-data (stride 1) from the blocked planes of dy"""
    B, H, W, Cin = xshape
    drop_amax(out)
    d = make_desc(xshape, ld_of(out), Cout, (Cout + 7) // 8 * 8, kh, kw, 1, pad, dil)
    with _Timed("split3", 0.0):
        wtp, wtsc = split2h_weight_t_blocked(w)
    with _Timed("dgrad_h2", 2.0 * d.B * d.Ho * d.Wo * Cout * Cin * kh * kw):
        check(lib.catseg_conv2d_bwd_data_f16x2_blocked(ctypes.byref(d), ptr(dyp), ptr(dysc), ptr(wtp), ptr(wtsc), ptr(out),
                                                       1 if accumulate else 0, stream()))
    return out


def bn_backward_pre(g, q, stats, gamma, partials, dgamma, dbeta, dq_out=None):
    """backward of relu(bn(q)) from the masked gradient g and the per-tile sums (partials, n_tiles) that conv_bwd_data(bn_src=...)
    returned: merge + apply pass only"""
    C = q.shape[-1]
    rows = rows_of(q)
    part, nt = partials[:2]
    if dq_out is None:
        dq_out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    ws = workspace(lib.catseg_bn_workspace(rows, C), q.device)
    with _Timed("hbm:bn_backward", 4.0 * q.numel() * 3):      # one pass: g and q read, dq written
        rec = new_amax(q.device) if _trunk_h2() else None
        check(lib.catseg_bn_backward_pre_amax(ptr(g), ld_of(g), ptr(q), ld_of(q), ptr(stats), ptr(gamma), ptr(part), nt, rows, C, ptr(dq_out),
                                              ld_of(dq_out), ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel(), ptr(rec), stream()))
    if rec is not None:
        dq_out._amax = rec
    return dq_out


def _bn_backward(dz, z, y, stats, gamma, relu, dgamma, dbeta, dres, dres_accumulate, dy_out, beta, rows, C, ws):
    rec = new_amax(y.device) if _trunk_h2() else None
    check(lib.catseg_bn_backward_amax(ptr(dz), ld_of(dz), ptr(z), ld_of(z) if z is not None else 0, ptr(y), ld_of(y), ptr(stats),
                                      ptr(gamma), ptr(beta), rows, C, 1 if relu else 0, ptr(dy_out), ld_of(dy_out), ptr(dgamma), ptr(dbeta),
                                      ptr(dres), ld_of(dres) if dres is not None else 0, 1 if dres_accumulate else 0, ptr(ws),
                                      ws.numel(), ptr(rec), stream()))
    if rec is not None:
        dy_out._amax = rec
    return dy_out


def nchw3_to_nhwc4(x):
    B, C, H, W = x.shape
    assert C == 3 and x.is_contiguous() and x.dtype == torch.float32
    y = torch.empty((B, H, W, 4), dtype=torch.float32, device=x.device)
    check(lib.catseg_nchw3_to_nhwc4(ptr(x), ptr(y), B, H, W, stream()))
    return y


def stem_pack_weight(w, O):
    pk = torch.empty((O, 7, 8, 4), dtype=torch.float32, device=w.device)
    check(lib.catseg_stem_pack_weight(ptr(w), ptr(pk), O, stream()))
    return pk


def stem_unpack_grad(pk, dw, O):
    check(lib.catseg_stem_unpack_grad(ptr(pk), ptr(dw), O, stream()))
    return dw


def axpy(src, dst, alpha=1.0, accumulate=True):
    drop_amax(dst)
    check(lib.catseg_axpy2d(ptr(src), ld_of(src), ptr(dst), ld_of(dst), rows_of(src), src.shape[-1], alpha,
                            1 if accumulate else 0, stream()))
    return dst


def scale_by_device_scalar(x, s):
    """x *= s (s: 0-dim / 1-element device tensor), in place"""
    assert x.is_contiguous() and s.numel() == 1 and s.dtype == torch.float32
    check(lib.catseg_scale_by_device_scalar(ptr(x), x.numel(), ptr(s), stream()))
    return x


def maxpool_fwd(x):
    B, H, W, C = x.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    check(lib.catseg_maxpool3x3s2_fwd(ptr(x), ld_of(x), ptr(y), C, ptr(idx), B, H, W, C, Ho, Wo, stream()))
    return y, idx


def maxpool_bwd(dy, idx, xshape):
    B, H, W, C = xshape
    dx = torch.empty(xshape, dtype=torch.float32, device=dy.device)
    check(lib.catseg_maxpool3x3s2_bwd(ptr(dy), ld_of(dy), ptr(idx), ptr(dx), C, B, H, W, C, dy.shape[1], dy.shape[2], stream()))
    return dx


def maxpool2_fwd(x):
    """F.max_pool2d(x, 2) (models/FCN.py:44-53 of the reference) -> (y, argmax position inside the window)"""
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, H // 2, W // 2, C), dtype=torch.uint8, device=x.device)
    check(lib.catseg_maxpool2x2_fwd(ptr(x), ld_of(x), ptr(y), C, ptr(idx), B, H, W, C, stream()))
    return y, idx


def maxpool2_bwd(dy, idx, out, accumulate=False):
    B, H, W, C = out.shape
    dst = torch.empty(out.shape, dtype=torch.float32, device=dy.device) if accumulate else out
    check(lib.catseg_maxpool2x2_bwd(ptr(dy), ld_of(dy), ptr(idx), ptr(dst), ld_of(dst), B, H, W, C, stream()))
    if accumulate:
        axpy(dst, out, 1.0, True)
    return out


def widen(t):
    """the [B, H, W, ld] tensor behind a class-logit view [B, H, W, K] whose rows are zero padded to ld floats (conv_fwd(zero_to=),
    engine.Ctx.dest): the 16-byte-granular kernels run on all ld columns"""
    B, H, W, _ = t.shape
    ld = ld_of(t)
    return t if ld == t.shape[-1] else torch.as_strided(t, (B, H, W, ld), (H * W * ld, W * ld, ld, 1))


def conv_transpose_fwd(x, w, bias, Cout, k, stride, pad, ld=32):
    """nn.ConvTranspose2d(Cin, Cout, k, stride, pad) on class-logit tensors (models/FCN.py:35-38, utils/torch_utils.py:151-168 of the
    reference): y = bias + backward-data of the convolution that has the same weight tensor (physical layout [Cin][k][k][Cout] = that
    convolution's OHWI).  x: [B, H, W, Cin] view with zero-padded rows of ld floats; returns the same kind of view of the output and the
    padded weight image the backward pass reuses."""
    B, H, W, Cin = x.shape
    Ho, Wo = (H - 1) * stride - 2 * pad + k, (W - 1) * stride - 2 * pad + k
    assert ld_of(x) == ld and Cout <= ld and Cin <= ld
    wp = weight_pad_cin(w, Cin, k * k, Cout, ld)
    y = new_act(B, Ho, Wo, Cout, x.device, ld=ld, zero=True)
    if bias is not None:
        check(lib.catseg_bias_rows(ptr(bias), ptr(y), ld, rows_of(y), Cout, stream()))
    conv_bwd_data(x, wp, (B, Ho, Wo, ld), k, k, stride, pad, 1, out=widen(y), accumulate=True)
    return y, wp


def conv_transpose_bwd(dy, x, wp, dw, dbias, k, stride, pad, dx, accumulate, ld=32):
    """gradients of conv_transpose_fwd: dw = backward-weight of the same convolution with the roles of input and output gradient swapped,
    dx = that convolution's FORWARD pass over dy"""
    Cin, Cout = x.shape[-1], dy.shape[-1]
    if ld_of(dy) != ld:     # the loss hands the gradient of the network's output over as a dense [B, H, W, K] tensor: re-pitch it (a copy)
        padded = new_act(dy.shape[0], dy.shape[1], dy.shape[2], Cout, dy.device, ld=ld, zero=True)
        padded.copy_(dy)
        dy = padded
    dyw = widen(dy)
    dwp = torch.empty_like(wp)
    conv_bwd_weight(dyw, x, dwp, None, k, k, stride, pad, 1)
    weight_unpad_cin(dwp, dw, Cin, k * k, Cout, ld)
    if dbias is not None:
        ws = workspace(256 * Cout * 4 + 1024, dy.device)
        check(lib.catseg_bias_grad(ptr(dy), ld_of(dy), rows_of(dy), Cout, ptr(dbias), ptr(ws), ws.numel(), stream()))
    if dx is not None:
        dst = new_act(x.shape[0], x.shape[1], x.shape[2], Cin, x.device, ld=ld, zero=True) if accumulate else dx
        conv_fwd(dyw, wp, None, Cin, k, k, stride, pad, 1, out=dst, zero_to=ld)
        if accumulate:
            axpy(widen(dst), widen(dx), 1.0, True)


def bilinear_fwd(x, Ho, Wo, align_corners, out=None, accumulate=False):
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
    with _Timed("hbm:bilinear_fwd", 4.0 * (x.numel() + out.numel() * (2 if accumulate else 1))):
        check(lib.catseg_bilinear_fwd(ptr(x), ld_of(x), ptr(out), ld_of(out), B, H, W, C, Ho, Wo, 1 if align_corners else 0,
                                      1 if accumulate else 0, stream()))
    if accumulate:
        drop_amax(out)
    elif amax_of(x) is not None:
        out._amax = amax_of(x)       # an interpolation with weights in [0, 1] that sum to 1: max|out| <= max|x| (the fuse sum's bound adds it)
    return out


def bilinear_bwd(dy, xshape, align_corners, out=None, zero_to=0, accumulate=False):
    B, H, W, C = xshape
    Ho, Wo = dy.shape[1], dy.shape[2]
    if out is None:
        out = new_act(B, H, W, C, dy.device, ld=max(zero_to, (C + 3) // 4 * 4))
        accumulate = False
    ws = workspace(B * H * Wo * C * 4, dy.device)
    with _Timed("hbm:bilinear_bwd", 4.0 * (dy.numel() + B * H * W * C * (2 if accumulate else 1))):
        _bilinear_bwd(dy, out, B, H, W, C, Ho, Wo, align_corners, zero_to, accumulate, ws)
    return out


def _bilinear_bwd(dy, out, B, H, W, C, Ho, Wo, align_corners, zero_to, accumulate, ws):
    check(lib.catseg_bilinear_bwd(ptr(dy), ld_of(dy), ptr(out), ld_of(out), B, H, W, C, Ho, Wo, 1 if align_corners else 0,
                                  zero_to, 1 if accumulate else 0, ptr(ws), ws.numel(), stream()))
    return out


def global_avgpool_fwd(x):
    B, H, W, C = x.shape
    y = torch.empty((B, 1, 1, C), dtype=torch.float32, device=x.device)
    check(lib.catseg_global_avgpool_fwd(ptr(x), ld_of(x), ptr(y), B, H * W, C, stream()))
    return y


def global_avgpool_bwd(dy, dx, accumulate):
    B, H, W, C = dx.shape
    check(lib.catseg_global_avgpool_bwd(ptr(dy), ptr(dx), ld_of(dx), B, H * W, C, 1 if accumulate else 0, stream()))
    return dx


def softmax_spatial_fwd(x, K):
    """x: [B, N, ld] buffer (logits in columns [0, K)); returns same-shape probabilities, pad columns zero."""
    B, N, ld = x.shape
    y = torch.empty_like(x)
    ws = workspace(lib.catseg_softmax_spatial_workspace(B, N), x.device)
    check(lib.catseg_softmax_spatial_fwd(ptr(x), ptr(y), B, N, K, ld, ptr(ws), ws.numel(), stream()))
    return y


def softmax_spatial_bwd(y, dy, dx, K, accumulate=False):
    B, N, ld = y.shape
    ws = workspace(lib.catseg_softmax_spatial_workspace(B, N), y.device)
    check(lib.catseg_softmax_spatial_bwd(ptr(y), ptr(dy), ptr(dx), B, N, K, ld, 1 if accumulate else 0, ptr(ws), ws.numel(), stream()))
    return dx


def softmax_rows_fwd(x, K, scale):
    rows, ld = x.shape
    y = torch.empty_like(x)
    check(lib.catseg_softmax_rows_fwd(ptr(x), ptr(y), rows, K, ld, scale, stream()))
    return y


def softmax_rows_bwd(y, dy, K, scale):
    rows, ld = y.shape
    dx = torch.empty_like(y)
    check(lib.catseg_softmax_rows_bwd(ptr(y), ptr(dy), ptr(dx), rows, K, ld, scale, stream()))
    return dx


def lovasz_softmax(logits, labels, weight=1.0, dlogits=None, accumulate=False, loss_out=None):
    """logits: [P, K] contiguous (NHWC flattened); labels int64 [P]."""
    Pn, K = logits.shape
    assert logits.is_contiguous() and labels.dtype == torch.int64 and labels.is_contiguous()
    if loss_out is None:
        loss_out = torch.empty(1, dtype=torch.float32, device=logits.device)
    ws = workspace(lib.catseg_lovasz_workspace(Pn, K), logits.device)
    # algorithmic minimum: logits read + gradient written (the sort passes on top are data dependent: active-set pruning)
    with _Timed("hbm:lovasz", 4.0 * Pn * K * (2 if dlogits is not None else 1) + 8.0 * Pn):
        _lovasz(logits, labels, Pn, K, weight, loss_out, dlogits, accumulate, ws)
    return loss_out


def _lovasz(logits, labels, Pn, K, weight, loss_out, dlogits, accumulate, ws):
    check(lib.catseg_lovasz_softmax(ptr(logits), ptr(labels), Pn, K, weight, ptr(loss_out), ptr(dlogits),
                                    1 if accumulate else 0, ptr(ws), ws.numel(), stream()))
    return loss_out


def lovasz_softmax_fwd(logits, labels, weight=1.0, want_grad=True):
    """forward half of lovasz_softmax for autograd: (loss, workspace) -- the workspace holds d loss / d prob for lovasz_softmax_bwd and is
    a buffer of its own (not the stream's shared scratch: other launches run between the two halves)"""
    Pn, K = logits.shape
    assert logits.is_contiguous() and labels.dtype == torch.int64 and labels.is_contiguous()
    loss_out = torch.empty(1, dtype=torch.float32, device=logits.device)
    need = lib.catseg_lovasz_workspace(Pn, K)
    ws = torch.empty(need, dtype=torch.uint8, device=logits.device) if want_grad else workspace(need, logits.device)
    with _Timed("hbm:lovasz", 4.0 * Pn * K * (2 if want_grad else 1) + 8.0 * Pn):
        check(lib.catseg_lovasz_softmax_fwd(ptr(logits), ptr(labels), Pn, K, weight, ptr(loss_out), 1 if want_grad else 0, ptr(ws), need, stream()))
    return loss_out, (ws if want_grad else None)


def lovasz_softmax_bwd(logits, ws, upstream, weight=1.0, dlogits=None, accumulate=False):
    """dlogits (+)= d loss / d logits * upstream (upstream: 1-element float32 device tensor, autograd's grad_output; None = 1)"""
    Pn, K = logits.shape
    if dlogits is None:
        dlogits = torch.empty_like(logits)
        accumulate = False
    assert upstream is None or (upstream.numel() == 1 and upstream.dtype == torch.float32 and upstream.is_cuda)
    with _Timed("hbm:lovasz", 0.0):
        check(lib.catseg_lovasz_softmax_bwd(ptr(logits), Pn, K, weight, ptr(upstream), ptr(dlogits), 1 if accumulate else 0, ptr(ws), ws.numel(), stream()))
    return dlogits


def cross_entropy(logits, labels, ignore_index, weight=1.0, dlogits=None, loss_out=None):
    Pn, K = logits.shape
    assert logits.is_contiguous() and labels.dtype == torch.int64 and labels.is_contiguous()
    if loss_out is None:
        loss_out = torch.empty(1, dtype=torch.float32, device=logits.device)
    ws = workspace(lib.catseg_ce_workspace(Pn), logits.device)
    check(lib.catseg_cross_entropy(ptr(logits), ptr(labels), Pn, K, ignore_index, weight, ptr(loss_out), ptr(dlogits),
                                   ptr(ws), ws.numel(), stream()))
    return loss_out


def ohem_cross_entropy(logits, labels, ignore_index, thresh, min_kept, weight=1.0, dlogits=None, loss_out=None):
    Pn, K = logits.shape
    assert logits.is_contiguous() and labels.dtype == torch.int64 and labels.is_contiguous()
    if loss_out is None:
        loss_out = torch.empty(1, dtype=torch.float32, device=logits.device)
    ws = workspace(lib.catseg_ohem_workspace(Pn), logits.device)
    check(lib.catseg_ohem_cross_entropy(ptr(logits), ptr(labels), Pn, K, ignore_index, thresh, min_kept, weight, ptr(loss_out),
                                        ptr(dlogits), ptr(ws), ws.numel(), stream()))
    return loss_out


def ingest_u8(img, lbl, lut=None, flips=None, pad_top=0, pad_bottom=0, mean=None, std=None, nhwc4=False):
    """img uint8 [B,H,W,3] and / or lbl uint8 [B,H,W] -> (x float32 [B,3,H',W] (or NHWC-4 [B,H',W,4]), labels int64 [B,H',W])"""
    ref = img if img is not None else lbl
    B, H, W = ref.shape[:3]
    Ho = H + pad_top + pad_bottom
    for t in (img, lbl, lut):
        assert t is None or (t.dtype == torch.uint8 and t.is_contiguous() and t.is_cuda)
    assert flips is None or (flips.dtype == torch.int32 and flips.numel() == B)
    x = x4 = labels = None
    if img is not None:
        assert img.shape == (B, H, W, 3)
        if nhwc4:
            x4 = torch.empty((B, Ho, W, 4), dtype=torch.float32, device=ref.device)
        else:
            x = torch.empty((B, 3, Ho, W), dtype=torch.float32, device=ref.device)
    if lbl is not None:
        assert lbl.shape == (B, H, W)
        labels = torch.empty((B, Ho, W), dtype=torch.int64, device=ref.device)
    check(lib.catseg_ingest_u8(ptr(img), ptr(lbl), B, H, W, ptr(lut), ptr(flips), pad_top, pad_bottom, ptr(mean), ptr(std),
                               ptr(x), ptr(x4), ptr(labels), stream()))
    return (x4 if nhwc4 else x), labels


def resize_nearest(src, Ho, Wo, flip=0, out=None, accumulate=False, divide_by=0.0):
    """NHWC nearest resize (F.interpolate(mode='nearest', size=(Ho, Wo))); flip 1 = flip the source, 2 = flip the result"""
    B, Hi, Wi, C = src.shape
    if out is None:
        out = new_act(B, Ho, Wo, C, src.device)
        accumulate = False
    check(lib.catseg_resize_nearest(ptr(src), ld_of(src), ptr(out), ld_of(out), B, Hi, Wi, Ho, Wo, C, flip, 1 if accumulate else 0,
                                    divide_by, stream()))
    return out


def confusion_matrix(logits, labels, cm=None):
    Pn, K = logits.shape
    if cm is None:
        cm = torch.zeros((K, K), dtype=torch.int32, device=logits.device)
    check(lib.catseg_confusion_matrix(ptr(logits), ptr(labels), Pn, K, ptr(cm), stream()))
    return cm


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    with _Timed("hbm:adam", 28.0 * p.numel()):          # 16 B read + 12 B written per parameter
        _adam_step(p, g, m, v, lr, step, beta1, beta2, eps, grad_scale)


def _adam_step(p, g, m, v, lr, step, beta1, beta2, eps, grad_scale):
    check(lib.catseg_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, stream()))


def adam_step_dev(p, g, m, v, hyper, beta1=0.9, beta2=0.999, eps=1e-8):
    """adam_step with {lr, bias corrections, gradient scale} in device memory (hyper: float32[4], optim.FusedAdam.upload_hyper)"""
    with _Timed("hbm:adam", 28.0 * p.numel()):
        check(lib.catseg_adam_step_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(hyper), beta1, beta2, eps, stream()))


def add_n_act(terms, relu, out=None):
    """out = act(sum(terms)); terms: up to 4 NHWC tensors of one shape (row strides may differ)"""
    t0 = terms[0]
    if out is None:
        out = torch.empty(t0.shape, dtype=torch.float32, device=t0.device)
    n = len(terms)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
    lds = (ctypes.c_int * n)(*[ld_of(t) for t in terms])
    if planes_ok(t0.shape[-1], rows_of(t0)) and all(amax_of(t) is not None for t in terms):
        rec = new_amax(t0.device)
        recs = (ctypes.c_void_p * n)(*[amax_of(t).data_ptr() for t in terms])
        buf = torch.empty(lib.catseg_planes_bytes(rows_of(t0), t0.shape[-1]), dtype=torch.uint8, device=t0.device)
        check(lib.catseg_add_n_act_planes(ptrs, lds, recs, n, ptr(out), ld_of(out), ptr(buf), rows_of(t0), t0.shape[-1], 1 if relu else 0, ptr(rec),
                                          stream()))
        out._amax = rec
        out._planes = Planes(buf, rec, t0.shape)
        return out
    rec = new_amax(t0.device) if _trunk_h2() else None
    check(lib.catseg_add_n_act_amax(ptrs, lds, n, ptr(out), ld_of(out), rows_of(t0), t0.shape[-1], 1 if relu else 0, ptr(rec), stream()))
    if rec is not None:
        out._amax = rec
    return out


def relu_bwd(dz, z):
    g = torch.empty(z.shape, dtype=torch.float32, device=z.device)
    check(lib.catseg_relu_bwd(ptr(dz), ld_of(dz), ptr(z), ld_of(z), ptr(g), ld_of(g), rows_of(z), z.shape[-1], stream()))
    return g


def weight_pad_cin(w, O, taps, cin, cpad):
    out = torch.empty((O, taps, cpad), dtype=torch.float32, device=w.device)
    check(lib.catseg_weight_pad_cin(ptr(w), ptr(out), O, taps, cin, cpad, 0, stream()))
    return out


def weight_unpad_cin(pk, dw, O, taps, cin, cpad):
    check(lib.catseg_weight_pad_cin(ptr(pk), ptr(dw), O, taps, cin, cpad, 1, stream()))
    return dw


def adaptive_avgpool_fwd(x, S):
    B, H, W, C = x.shape
    y = torch.empty((B, S, S, C), dtype=torch.float32, device=x.device)
    check(lib.catseg_adaptive_avgpool_fwd(ptr(x), ld_of(x), ptr(y), B, H, W, C, S, stream()))
    return y


def adaptive_avgpool_bwd(dy, dx, S, accumulate):
    B, H, W, C = dx.shape
    check(lib.catseg_adaptive_avgpool_bwd(ptr(dy), ptr(dx), ld_of(dx), B, H, W, C, S, 1 if accumulate else 0, stream()))
    return dx


def fold_bn(w, bias, gamma, beta, rm, rv, eps, O, per_out):
    """conv weight / bias with an eval-mode BatchNorm folded in (inference fast path)"""
    wf = torch.empty(O * per_out, dtype=torch.float32, device=w.device)
    bf = torch.empty(O, dtype=torch.float32, device=w.device)
    check(lib.catseg_fold_bn(ptr(w), ptr(bias), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), eps, O, per_out, ptr(wf), ptr(bf), stream()))
    return wf, bf


def conv_fwd_fused(x, w, bias, residual, relu, Cout, kh, kw, stride=1, pad=0, dil=1, out=None, stem4=False, groups=1):
    B, H, W, Cin = x.shape
    Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
    if out is None:
        out = new_act(B, Ho, Wo, Cout, x.device)
    flops = 2.0 * B * Ho * Wo * Cout * (3 if stem4 else Cin // groups) * kh * kw
    if ("fwd" in B3_OPS and not stem4 and groups == 1 and w.numel() == Cout * kh * kw * Cin and _b3_eligible(B * Ho * Wo, Cout, kh * kw, Cin)
            and B3_BLOCKED and Cout > 192 and Cin % 16 == 0 and 6 * Cout * kh * kw * Cin < (1 << 32) - 64):
        # the bf16x3 kernel with the fused epilogue; the batch is cut so that the three blocked planes of a piece stay below 4 GB
        # (UPerNet's 3x3 2048 -> 512 on a 4 x 272 x 480 map: 6.4 GB of planes in one piece)
        per_img = 6 * H * W * Cin
        nb = max(1, min(B, B3_PLANE_LIMIT // per_img))
        if per_img < B3_PLANE_LIMIT:
            wp = None
            for b0 in range(0, B, nb):
                xs, os_ = (x, out) if nb == B else (x[b0:b0 + nb], out[b0:b0 + nb])    # (x itself: the split-plane cache is keyed by identity)
                rs = residual[b0:b0 + nb] if residual is not None else None
                d = make_desc(xs.shape, Cin, Cout, ld_of(out), kh, kw, stride, pad, dil)
                if _h2():
                    with _Timed("split3", 0.0):
                        xp, xsc = _split3_cached(xs, "h2") if nb == B else split2h_blocked(xs)
                        if wp is None:      # (w: OHWI weights, 4-D or the flat buffer catseg_fold_bn writes)
                            wp = torch.empty((2, kh * kw * Cin // 16, Cout, 16), dtype=torch.int16, device=w.device)
                            wsc = torch.empty(2, dtype=torch.int32, device=w.device)
                            check(lib.catseg_split2h_weight_blocked(ptr(w), Cout, kh * kw, Cin, ptr(wp), ptr(wsc), stream()))
                    with _Timed("fwd_h2", flops * xs.shape[0] / B):
                        check(lib.catseg_conv2d_fwd_fused_f16x2_blocked(ctypes.byref(d), ptr(xp), ptr(xsc), ptr(wp), ptr(wsc), ptr(bias), ptr(rs),
                                                                        ld_of(residual) if residual is not None else 0, 1 if relu else 0,
                                                                        ptr(os_), stream()))
                    continue
                with _Timed("split3", 0.0):
                    xp = _split3_cached(xs, "blk") if nb == B else split3_blocked(xs)[0]
                    if wp is None:      # (w: OHWI weights, 4-D or the flat buffer catseg_fold_bn writes)
                        wp = torch.empty((3, kh * kw * Cin // 16, Cout, 16), dtype=torch.int16, device=w.device)
                        check(lib.catseg_split3_weight_blocked(ptr(w), Cout, kh * kw, Cin, ptr(wp), stream()))
                with _Timed("fwd_b3", flops * xs.shape[0] / B):
                    check(lib.catseg_conv2d_fwd_fused_bf16x3_blocked(ctypes.byref(d), ptr(xp), ptr(wp), ptr(bias), ptr(rs),
                                                                     ld_of(residual) if residual is not None else 0, 1 if relu else 0, ptr(os_), stream()))
            return out
    d = make_desc(x.shape, ld_of(x), Cout, ld_of(out), kh, kw, stride, pad, dil, stem4, groups)
    with _Timed("fwd", flops):
        check(lib.catseg_conv2d_fwd_fused(ctypes.byref(d), ptr(x), ptr(w), ptr(bias), ptr(residual),
                                          ld_of(residual) if residual is not None else 0, 1 if relu else 0, ptr(out), stream()))
    return out


# ---------------------------------------------------------------------------------------------- split precision (bf16 x 3)
def split3(x):
    """fp32 NHWC activation [..., C] (pixel stride ld) -> int16 planes [3, rows, roundup(C, 8)]"""
    rows, C, ld = rows_of(x), x.shape[-1], ld_of(x)
    planes = torch.empty((3, rows, (C + 7) // 8 * 8), dtype=torch.int16, device=x.device)
    check(lib.catseg_split3(ptr(x), ld, rows, C, ptr(planes), stream()))
    return planes


def split3_blocked(x, with_planar=False):
    """fp32 NHWC activation -> (blocked planes [3, ceil(C/16), rows, 16], planar planes [3, rows, roundup(C, 8)] or None), one pass"""
    rows, C, ld = rows_of(x), x.shape[-1], ld_of(x)
    blk = torch.empty((3, (C + 15) // 16, rows, 16), dtype=torch.int16, device=x.device)
    planar = torch.empty((3, rows, (C + 7) // 8 * 8), dtype=torch.int16, device=x.device) if with_planar else None
    check(lib.catseg_split3_blocked(ptr(x), rows, C, ld, ptr(blk), ptr(planar), stream()))
    return blk, planar


def split2h(x, blocked=True, planar=False):
    """fp32 NHWC activation -> fp16 planes of x * 2^e in the blocked layout [2, ceil(C/16), rows, 16] and / or the planar layout
    [2, rows, roundup(C, 8)] (one pass for both) + the device record int32[2] = {bits of max|x|, e}: (blocked or None, planar or None, scale)"""
    rows, C, ld = rows_of(x), x.shape[-1], ld_of(x)
    blk = torch.empty((2, (C + 15) // 16, rows, 16), dtype=torch.int16, device=x.device) if blocked else None
    pl = torch.empty((2, rows, (C + 7) // 8 * 8), dtype=torch.int16, device=x.device) if planar else None
    scale = torch.empty(2, dtype=torch.int32, device=x.device)
    recs = amax_records_of(x) if SPLIT_BOUND else None
    if recs:        # the producers of x left max|x|: no pass over x to find it (csrc/igemm_f16x2.hip: catseg_split2h_bound)
        r = list(recs) + [None] * (4 - len(recs))
        check(lib.catseg_split2h_bound(ptr(x), rows, C, ld, ptr(blk), ptr(pl), ptr(scale), ptr(r[0]), ptr(r[1]), ptr(r[2]), ptr(r[3]), stream()))
    else:
        check(lib.catseg_split2h(ptr(x), rows, C, ld, ptr(blk), ptr(pl), ptr(scale), stream()))
    return blk, pl, scale


def split2h_blocked(x):
    blk, _, scale = split2h(x, True, False)
    return blk, scale


# Weight images of the head layers (forward: blocked planes; backward-data: the transposed bank's).  Each was amax + split launched in line, in
# front of its GEMM on the strictly serial head path (~55 us with the launch gaps, six times per step).  Round 6: a layer that took this route
# once keeps its two image buffers, and while a step is being captured into a hipGraph EngineNet._run refreshes them all on the weight-image
# side stream beside stem + stage 1 (h2_weight_images_refresh: the same hand-over as the trunk's banks, ops.images_ready); everywhere else the
# images are made in line as before.
H2W_BANK = _plan.get("h2w_bank")
_h2w_bank = None      # the running network's bank (EngineNet._run sets it: the buffers live as long as the network):
                      # (data_ptr, shape, transposed) -> {"w": weights, "planes", "scale", "fresh": made for the CURRENT step by the refresh}


def _h2w_entry(w, transposed):
    O, I, kh, kw = w.shape
    key = (w.data_ptr(), tuple(w.shape), transposed)
    e = _h2w_bank.get(key) if _h2w_bank is not None else None
    if e is None:
        rows = kh * kw * ((O + 15) // 16) if transposed else kh * kw * I // 16
        e = {"w": w, "planes": torch.empty((2, rows, I if transposed else O, 16), dtype=torch.int16, device=w.device),
             "scale": torch.empty(2, dtype=torch.int32, device=w.device), "fresh": False}
        if H2W_BANK and _h2w_bank is not None and len(_h2w_bank) < 64:
            _h2w_bank[key] = e
    return e


def _h2w_launch(e, transposed):
    w = e["w"]
    O, I, kh, kw = w.shape
    fn = lib.catseg_split2h_weight_t_blocked if transposed else lib.catseg_split2h_weight_blocked
    check(fn(ptr(w), O, kh * kw, I, ptr(e["planes"]), ptr(e["scale"]), stream()))


def h2_weight_images_begin(bank):
    """a network's pass starts: its bank becomes the current one, nothing in it is valid for this step yet"""
    global _h2w_bank
    _h2w_bank = bank
    for e in bank.values():
        e["fresh"] = False


def h2_weight_images_refresh():
    """(on the weight-image side stream) both images of every head layer seen so far, for the step that starts now"""
    for (_, _, transposed), e in _h2w_bank.items():
        _h2w_launch(e, transposed)
        e["fresh"] = True


def split2h_weight_blocked(w):
    """[O, I, kh, kw] weights (physical OHWI, I % 16 == 0) -> (fp16 planes [2, kh*kw*I/16, O, 16], scale record): forward operand"""
    e = _h2w_entry(w, False)
    if e["fresh"]:
        images_ready()
    else:
        _h2w_launch(e, False)
    return e["planes"], e["scale"]


def split2h_weight_t_blocked(w):
    """OHWI weights -> (fp16 planes of the transposed bank [2, taps*roundup(O,16)/16, Cin, 16], scale record): backward-data operand"""
    e = _h2w_entry(w, True)
    if e["fresh"]:
        images_ready()
    else:
        _h2w_launch(e, True)
    return e["planes"], e["scale"]


def split3_weight_blocked(w):
    """[O, I, kh, kw] weights (physical OHWI, I % 16 == 0) -> blocked planes [3, kh*kw*I/16, O, 16] (forward operand)"""
    O, I, kh, kw = w.shape
    planes = torch.empty((3, kh * kw * I // 16, O, 16), dtype=torch.int16, device=w.device)
    check(lib.catseg_split3_weight_blocked(ptr(w), O, kh * kw, I, ptr(planes), stream()))
    return planes


def split3_weight_t_blocked(w):
    """OHWI weights -> blocked planes of the transposed bank [3, taps*roundup(O,16)/16, Cin, 16] (backward-data operand)"""
    O, Cin, kh, kw = w.shape
    planes = torch.empty((3, kh * kw * ((O + 15) // 16), Cin, 16), dtype=torch.int16, device=w.device)
    check(lib.catseg_split3_weight_t_blocked(ptr(w), O, kh * kw, Cin, ptr(planes), stream()))
    return planes


def split3_weight(w):
    """[O, I, kh, kw] weights (physical OHWI) -> planes [3, O, kh*kw*I] (forward operand; I % 8 == 0)"""
    O, I, kh, kw = w.shape
    planes = torch.empty((3, O, kh * kw * I), dtype=torch.int16, device=w.device)
    check(lib.catseg_split3(ptr(w), kh * kw * I, O, kh * kw * I, ptr(planes), stream()))
    return planes


def split3_weight_t(w):
    """OHWI weights -> planes of the transposed bank [3, Cin, taps, roundup(O, 8)] (backward-data operand)"""
    O, Cin, kh, kw = w.shape
    planes = torch.empty((3, Cin, kh * kw, (O + 7) // 8 * 8), dtype=torch.int16, device=w.device)
    check(lib.catseg_split3_weight_t(ptr(w), O, kh * kw, Cin, ptr(planes), stream()))
    return planes


def conv_fwd_b3(xshape, xp, wp, bias, Cout, kh, kw, stride=1, pad=0, dil=1, out=None, zero_to=0):
    B, H, W, Cin = xshape
    Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
    if out is None:
        out = new_act(B, Ho, Wo, Cout, xp.device, ld=max(zero_to, (Cout + 3) // 4 * 4))
    d = make_desc(xshape, Cin, Cout, ld_of(out), kh, kw, stride, pad, dil)
    with _Timed("fwd", 2.0 * B * Ho * Wo * Cout * Cin * kh * kw):
        check(lib.catseg_conv2d_fwd_bf16x3(ctypes.byref(d), ptr(xp), ptr(wp), ptr(bias), ptr(out), zero_to, stream()))
    return out


def conv_bwd_data_b3(dyp, wtp, xshape, Cout, kh, kw, stride=1, pad=0, dil=1, out=None, accumulate=False):
    B, H, W, Cin = xshape
    if out is None:
        out = new_act(B, H, W, Cin, dyp.device)
        accumulate = False
    d = make_desc(xshape, ld_of(out), Cout, (Cout + 7) // 8 * 8, kh, kw, stride, pad, dil)
    with _Timed("dgrad", 2.0 * B * d.Ho * d.Wo * Cout * Cin * kh * kw):
        check(lib.catseg_conv2d_bwd_data_bf16x3(ctypes.byref(d), ptr(dyp), ptr(wtp), ptr(out), 1 if accumulate else 0, stream()))
    return out


def conv_fwd_b3_blocked(xshape, xblk, wblk, bias, Cout, kh, kw, stride=1, pad=0, dil=1, out=None, zero_to=0):
    """forward from BLOCKED planes (split3_blocked / split3_weight_blocked): always the 256 x 256 register-pipelined kernel"""
    B, H, W, Cin = xshape
    Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
    if out is None:
        out = new_act(B, Ho, Wo, Cout, xblk.device, ld=max(zero_to, (Cout + 3) // 4 * 4))
    d = make_desc(xshape, Cin, Cout, ld_of(out), kh, kw, stride, pad, dil)
    with _Timed("fwd", 2.0 * B * Ho * Wo * Cout * Cin * kh * kw):
        check(lib.catseg_conv2d_fwd_bf16x3_blocked(ctypes.byref(d), ptr(xblk), ptr(wblk), ptr(bias), ptr(out), zero_to, None, 0, None, None, stream()))
    return out


def conv_bwd_data_b3_blocked(dyblk, wtblk, xshape, Cout, kh, kw, stride=1, pad=0, dil=1, out=None, accumulate=False):
    B, H, W, Cin = xshape
    if out is None:
        out = new_act(B, H, W, Cin, dyblk.device)
        accumulate = False
    d = make_desc(xshape, ld_of(out), Cout, (Cout + 7) // 8 * 8, kh, kw, stride, pad, dil)
    with _Timed("dgrad", 2.0 * B * d.Ho * d.Wo * Cout * Cin * kh * kw):
        check(lib.catseg_conv2d_bwd_data_bf16x3_blocked(ctypes.byref(d), ptr(dyblk), ptr(wtblk), ptr(out), 1 if accumulate else 0, stream()))
    return out


# ---------------------------------------------------------------------------------------------- uint8 augmentation (csrc/augment.hip)
def aug_pad_flip_u8(img, flips, pad_top, pad_bottom):
    """uint8 [B,H,W,C] -> uint8 [B,H+pad,W,C]: FlipNP flags (bit 0 horizontal, bit 1 vertical) + PadNP reflect rows"""
    assert img.dtype == torch.uint8 and img.is_contiguous() and img.is_cuda and img.dim() == 4
    B, H, W, C = img.shape
    out = torch.empty((B, H + pad_top + pad_bottom, W, C), dtype=torch.uint8, device=img.device)
    check(lib.catseg_aug_pad_flip_u8(ptr(img), ptr(out), B, H, W, C, ptr(flips), pad_top, pad_bottom, stream()))
    return out


def aug_gaussian_blur(img, radius, ww, fw):
    """ImageFilter.GaussianBlur per image: three box passes along W, then three along H; radius[b] < 0: image unchanged"""
    assert img.dtype == torch.uint8 and img.is_contiguous() and img.is_cuda and img.shape[-1] == 3
    B, H, W, _ = img.shape
    a, b = img, torch.empty_like(img)
    for axis in (1, 1, 1, 0, 0, 0):
        check(lib.catseg_aug_box_blur(ptr(a), ptr(b), B, H, W, axis, ptr(radius), ptr(ww), ptr(fw), stream()))
        a, b = b, (torch.empty_like(img) if a is img else a)
    return a


def aug_color_op(img, op, factor):
    """one torchvision ColorJitter operation per image, IN PLACE (op 0 brightness, 1 contrast, 2 saturation, 3 hue with factor = H shift)"""
    assert img.dtype == torch.uint8 and img.is_contiguous() and img.is_cuda and img.shape[-1] == 3
    B, H, W, _ = img.shape
    ws = workspace(8 * B + 256, img.device)
    check(lib.catseg_aug_color_op(ptr(img), B, H, W, ptr(op), ptr(factor), ptr(ws), ws.numel(), stream()))
    return img
