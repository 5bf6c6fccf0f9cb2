"""The execution plan: ONE registry of every route / threshold switch of the host layer.

Until round 5 the routes were selected by ~50 ad-hoc `os.environ.get("CATSEG_...")` reads spread over ops.py, engine.py, dist.py and the
models; a bench line could not say which plan it had run.  Now every switch is a FIELD of this registry -- name, default, parser, one line of
documentation, and the module attribute that holds its live value (tests and A/B tools flip the attribute; nothing else reads the
environment):

    plan.get("heads")            the start-up value: CATSEG_PLAN="heads=bf16x3,precision=fp32" > CATSEG_HEADS (the field's own variable) > default
    plan.active()                {field: live value} read back from the owning modules: what `ops.plan()` returns and bench.py prints as
                                 `config.plan`
    plan.describe()              the documentation table

Variants that were measured and NOT adopted (asynchronous head backward-weight on an extra stream, stream-skew / branch-priority knobs of
the parallel regions, residual-gradient fusion into backward-data epilogues, fuse-chain placement on the destination branch) are no longer
in the product path; their measurements are in docs/DESIGN_history.md and profiles/.
"""
import os


def _bool(v):
    return str(v).strip().lower() not in ("0", "false", "no", "off", "")


def _csv_str(v):
    return v if isinstance(v, tuple) else tuple(s for s in str(v).split(",") if s)


def _csv_int(v):
    return v if isinstance(v, tuple) else tuple(int(s) for s in str(v).split(",") if s)


# name: (default, parser, owner "module:ATTRIBUTE" (or None: read through get() where it is used), documentation)
FIELDS = {
    # ---- arithmetic of the convolution routes
    "precision": ("bf16x3", str, "ops:PRECISION", "split-precision routes on ('bf16x3') or exact fp32 MFMA chains everywhere ('fp32')"),
    "heads": ("f16x2", str, "ops:HEADS", "large layers (K >= 2048, >= 192 columns): two fp16 planes / 3 products ('f16x2') or three bf16 planes / 6 ('bf16x3')"),
    "trunk": ("f16x2", str, "ops:TRUNK", "direct 3x3 kernels of the HRNet widths: 'f16x2' or 'bf16x3'"),
    "trunk_planes": (True, _bool, "ops:PLANES", "trunk tensors as producer-written fp16 x 2 planes (csrc/dconv3_pl.hip); off: split inside the convolution kernels"),
    "planes_widths": ((96, 192, 384), _csv_int, "ops:PLANES_WIDTHS", "channel counts that take the planes route"),
    "h2t": ("blocked", str, None, "backward-weight of the head layers reads the blocked planes of forward / backward-data ('blocked') or its own planar set ('planar')"),
    "split_bound": (True, _bool, "ops:SPLIT_BOUND", "f16x2 split passes take max|x| from the producers' amax records instead of an amax pass"),
    "p1": (True, _bool, "ops:P1", "1 x 1 layers on csrc/pconv1.hip (split in registers)"),
    "p1_ops": (("fwd", "dgrad", "wgrad"), _csv_str, "ops:P1_OPS", "directions of the pointwise route"),
    "wide_1x1_min_rows": (60000, int, "ops:B3_1X1_MIN_ROWS", "1 x 1 layers with both extents >= 512 take the blocked-plane kernels (split pass + igemm_f16x2) from this many pixels; below: the pointwise route"),
    "wide_1x1_min_dim": (512, int, "ops:B3_1X1_MIN_DIM", "... and both extents >= this (their product >= 1024 x this)"),
    "p1_min_rows": (60000, int, "ops:P1_MIN_ROWS", "pointwise route: minimum output pixels"),
    "p1_wgrad_min_dim": (64, int, "ops:P1_WGRAD_MIN_DIM", "pointwise route, backward-weight: minimum of (Cin, Cout)"),
    "g1": (True, _bool, "ops:G1", "strided / non-square 3 x 3 layers as gather launches of csrc/pconv1.hip"),
    "g1_min_rows": (4096, int, "ops:G1_MIN_ROWS", "gather route: minimum output pixels"),
    "g1_dgrad_min_cin": (48, int, "ops:G1_DGRAD_MIN_CIN", "gather route, backward-data: minimum input channels"),
    "g1_min_cin": (0, int, "ops:G1_MIN_CIN", "gather route: minimum input channels"),
    "g1_ops": (("fwd", "dgrad", "wgrad"), _csv_str, "ops:G1_OPS", "directions of the gather route"),
    "stem3": (True, _bool, "ops:STEM3", "HRNet stem conv1 on the direct fp64-accumulating kernels (csrc/stem3.hip)"),
    "stem7": (True, _bool, "ops:STEM7", "ResNet stem conv1, training forward, on the direct fp64-accumulating kernel (csrc/stem7.hip)"),
    "exact_early": ("all", str, None, "HRNet layers whose forward stays on exact fp32 operands: 'layer1' / 'stem' / 'all' (models/HRNetv2.py)"),
    "gemm_tn_split": (True, _bool, "ops:GEMM_TN_SPLIT", "OCR proxy / key / value reductions over all pixels as row-chunk GEMMs + fixed-order slab sum"),
    "relu_bits": (True, _bool, "ops:RELU_BITS", "ReLU masks of residual blocks as bits for the BatchNorm backward"),
    "concat_planes": (True, _bool, "ops:CONCAT_PLANES", "HRNet head input written only as blocked f16x2 planes"),
    "head_dy_planes": (True, _bool, "ops:HEAD_DY_PLANES", "BatchNorm backward of the head layers writes dy only as blocked planes"),
    "head_fuse": (True, _bool, "ops:HEAD_FUSE", "the K-class classifier behind a head's BatchNorm + ReLU fused with it (csrc/headfuse.h): the normalised activation and its gradient are never written"),
    "h2w_bank": (True, _bool, "ops:H2W_BANK", "the head layers' weight images keep their buffers and are refreshed on the weight-image side stream at the start of a captured step (in line otherwise)"),
    # ---- launch-shape knobs of the library (include/catseg_debug.h; 0 = the library's default)
    "wg_blocks": (0, int, None, "blocks of the direct backward-weight kernel csrc/dwgrad3_b3.hip"),
    "dc_blocks": (0, int, None, "persistent blocks of the direct 3x3 kernel csrc/dconv3_b3.hip"),
    "pl_slots": (0, int, None, "persistent blocks of the planes kernel csrc/dconv3_pl.hip"),
    "pl_pair": (0, int, None, "bit mask of channel counts whose planes kernel runs two tiles per block"),
    "wp96_blocks": (0, int, None, "blocks of the 96+ channel backward-weight kernel on planes"),
    "igemm_splits": (0, int, None, "forced split count of the fp32 implicit-GEMM backward-weight"),
    "h2w_slow_epilogue": (0, int, None, "1: igemm_h2w8_kernel always runs its general (element-predicated) epilogue (A/B of the full-row-tile epilogue)"),
    "h2w_waves": (0, int, None, "waves per block of the head layers' forward / backward-data kernel: 8 (default) or 4"),
    # ---- execution
    "branch_streams": (4, int, "engine:BRANCH_STREAMS", "HIP streams a parallel region spreads its branches over"),
    "last_branch_on_main": (True, _bool, "engine:LAST_BRANCH_ON_MAIN", "the last branch of a four-branch region runs on the launch stream"),
    "prep_async": (True, _bool, "engine:PREP_ASYNC", "per-step weight images on a side stream beside stem + stage 1 (while capturing a hipGraph)"),
    # ---- data parallel
    "bucket_mb": (32.0, float, None, "gradient bucket size of the all-reduce (dist.GradSync)"),
    "tail_bucket_mb": (4.0, float, None, "size of the bucket whose gradients are ready last (stem)"),
    "segment_mb": (48.0, float, None, "hipGraph replay, data parallel: gradient MB released before the captured step is cut (graph.GraphedTrainStep)"),
    "comm_priority": ("high", str, None, "priority of RCCL's communicator stream ('high' / 'normal')"),
}

_ENV_NAME = {"trunk_planes": "CATSEG_TRUNK_PLANES"}     # (every other field: CATSEG_<NAME>)
LIBRARY_KNOBS = {"wg_blocks": "catseg_debug_set_dwgrad3_blocks", "dc_blocks": "catseg_debug_set_dconv3_blocks",
                 "pl_slots": "catseg_debug_set_dconv3_pl_slots", "pl_pair": "catseg_debug_set_dconv3_pl_pair",
                 "wp96_blocks": "catseg_debug_set_dwgrad3_pl_blocks", "igemm_splits": "catseg_debug_set_splits",
                 "h2w_waves": "catseg_debug_set_h2w_waves", "h2w_slow_epilogue": "catseg_debug_set_h2w_slow_epilogue"}
_plan_env = None


def _plan_overrides():
    global _plan_env
    if _plan_env is None:
        _plan_env = {}
        for item in os.environ.get("CATSEG_PLAN", "").replace(";", ",").split(","):
            # values that are lists use '+' inside CATSEG_PLAN: planes_widths=96+192
            if "=" in item:
                k, v = item.split("=", 1)
                if k.strip() not in FIELDS:
                    raise ValueError("CATSEG_PLAN: unknown field %r (fields: %s)" % (k.strip(), ", ".join(sorted(FIELDS))))
                _plan_env[k.strip()] = v.strip().replace("+", ",")
    return _plan_env


def get(name):
    """start-up value of a field: CATSEG_PLAN entry > the field's own environment variable > default"""
    default, parse, _, _ = FIELDS[name]
    ov = _plan_overrides()
    if name in ov:
        return parse(ov[name])
    env = os.environ.get(_ENV_NAME.get(name, "CATSEG_" + name.upper()))
    if env is not None:
        return parse(env)
    return default


def active():
    """{field: live value}: module attributes are read back (tests and A/B tools change them after import)"""
    import importlib
    out = {}
    for name, (_, _, owner, _) in FIELDS.items():
        if owner is None:
            v = get(name)
            if name == "h2t":
                v = "blocked" if importlib.import_module(__package__ + ".ops").H2T_BLOCKED else "planar"
        else:
            mod, attr = owner.split(":")
            v = getattr(importlib.import_module(__package__ + "." + mod), attr)
        out[name] = list(v) if isinstance(v, tuple) else v
    return out


def non_default():
    """the fields whose live value differs from the default (what a bench line needs to say beyond 'default plan')"""
    act = active()
    return {k: v for k, v in act.items() if (list(FIELDS[k][0]) if isinstance(FIELDS[k][0], tuple) else FIELDS[k][0]) != v}


def describe():
    return "\n".join("%-20s default %-24r %s" % (k, v[0], v[3]) for k, v in FIELDS.items())
