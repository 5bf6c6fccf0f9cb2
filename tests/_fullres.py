"""Shared machinery of the full-resolution parity tests (tests/test_fullres_gpu.py) and of tools/parity_fullres.py: ONE evaluation of the
CPU oracle per network (fp32 train step + fp64 forward), the HIP path under several arithmetic plans, and a JSON record of every figure so
that a PASSING run leaves evidence (profiles/r04_parity_fullres.json is a copy of the file this writes).

Test infrastructure: the oracle is only the checker here; every HIP call goes through the C ABI."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.environ.get("CATSEG_PARITY_RECORD", os.path.join(ROOT, "gpurun_out", "parity_fullres.json"))

# arithmetic plans: name -> (ops.PRECISION, ops.TRUNK, ops.HEADS, ops.PLANES)
PLANS = {
    "production": ("bf16x3", "f16x2", "f16x2", True),            # trunk on producer-written fp16 x 2 planes
    "trunk_in_kernel_split": ("bf16x3", "f16x2", "f16x2", False),  # the round-3 route: fp32 tensors, split inside the convolution kernels
    "fp32": ("fp32", "f16x2", "f16x2", True),
    "trunk_bf16x3": ("bf16x3", "bf16x3", "f16x2", True),
    "heads_bf16x3": ("bf16x3", "f16x2", "bf16x3", True),
    "all_bf16x3": ("bf16x3", "bf16x3", "bf16x3", True),
}


def record(section, key, values):
    """merge {section: {key: values}} into the JSON record (one file for the whole run)"""
    os.makedirs(os.path.dirname(RECORD), exist_ok=True)
    data = {}
    if os.path.exists(RECORD):
        try:
            with open(RECORD) as f:
                data = json.load(f)
        except ValueError:
            data = {}
    data.setdefault(section, {})[key] = values
    with open(RECORD, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def block_labels(B, H, W, K, seed):
    g = torch.Generator().manual_seed(seed)
    lbl = torch.randint(0, K + 1, (B, H // 32, W // 32), generator=g)
    return lbl.repeat_interleave(32, 1).repeat_interleave(32, 2).contiguous()


class set_plan:
    """context: run the HIP path under one arithmetic plan (thresholds untouched: the production layer selection)"""

    def __init__(self, name, batch=8):
        """batch: the batch the plan is run at.  The pixel-count thresholds of the pointwise / gather routes (ops.P1_MIN_ROWS, ops.G1_MIN_ROWS) are
        stated for the bench's batch of 8; a parity run at batch 2 scales them by 2 / 8 so that it exercises the SAME layer selection -- with
        the thresholds untouched the 65 280-pixel maps of a batch-2 step would stay on the fp32 kernels and the test would pass without ever
        running csrc/pconv1.hip on the layers the benchmark runs it on."""
        self.name, self.batch = name, batch

    def __enter__(self):
        from miccai2021_cataract_semantic_segmentation_amd import ops
        self.saved = (ops.PRECISION, ops.TRUNK, ops.HEADS, ops.PLANES, ops.P1_MIN_ROWS, ops.G1_MIN_ROWS)
        ops.release_b3_cache()
        ops.PRECISION, ops.TRUNK, ops.HEADS, ops.PLANES = PLANS[self.name]
        ops.P1_MIN_ROWS, ops.G1_MIN_ROWS = ops.P1_MIN_ROWS * self.batch // 8, ops.G1_MIN_ROWS * self.batch // 8
        return self

    def __exit__(self, *exc):
        from miccai2021_cataract_semantic_segmentation_amd import ops
        ops.PRECISION, ops.TRUNK, ops.HEADS, ops.PLANES, ops.P1_MIN_ROWS, ops.G1_MIN_ROWS = self.saved
        ops.release_b3_cache()
        return False


def argmax_figures(hip, cpu32, cpu64):
    """label-map figures against the fp64 forward: unmasked disagreement counts, and the count of pixels where HIP differs from fp64
    although the fp64 top-2 margin exceeds 2.2 x the logit error made (must be 0: every disagreement is a tie broken the other way)"""
    a_h, a_c, a_64 = hip.argmax(1), cpu32.argmax(1), cpu64.argmax(1)
    top2 = cpu64.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    err = float((hip.double() - cpu64).abs().max())
    return {"pixels": int(a_64.numel()), "hip_vs_cpu32": int((a_h != a_c).sum()), "hip_vs_fp64": int((a_h != a_64).sum()),
            "cpu32_vs_fp64": int((a_c != a_64).sum()), "outside_error_band": int(((a_h != a_64) & (margin > 2.2 * err)).sum())}


_oracle_cache = {}


def hrnet48_oracle(B=2, H=544, W=960, K=25, wseed=41, xseed=9, lseed=10):
    """the bench model's train step on the CPU oracle (fp32, with gradients) and its fp64 forward, evaluated once per process"""
    key = ("hrnet48", B, H, W, wseed, xseed, lseed)
    if key in _oracle_cache:
        return _oracle_cache[key]
    import bench
    from oracle import nets as ON, losses as OL
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    cfg = dict(bench.MODELS["ocrnet_hrnet48"][0])
    spec = spec_of(OCRNet(dict(cfg), 3).state_dict())
    S = fill_state(spec, wseed)
    g = torch.Generator().manual_seed(xseed)
    x = torch.rand(B, 3, H, W, generator=g)
    lbl = block_labels(B, H, W, K, lseed)
    for k, v in S.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_()
    oi, of = ON.ocrnet_hrnet_forward(S, x, train=True)
    ol = OL.two_scale_lovasz(oi, of, lbl, 0.4, 1.0)
    ol.backward()
    S64 = {k: (v.detach().double() if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, wseed).items()}
    with torch.no_grad():
        i64, f64 = ON.ocrnet_hrnet_forward(S64, x.double(), train=True)
        l64 = float(OL.two_scale_lovasz(i64, f64, lbl, 0.4, 1.0))
    res = dict(cfg=cfg, spec=spec, S=S, x=x, lbl=lbl, interm32=oi.detach(), final32=of.detach(), loss32=float(ol), final64=f64, interm64=i64,
               loss64=l64, wseed=wseed)
    _oracle_cache[key] = res
    return res


def hrnet48_hip(orc, plan):
    """one train step of the bench model on the HIP path under `plan`, from the oracle's weights and inputs: (model, interm, final, loss, kinds)"""
    from oracle.state import fill_state
    from miccai2021_cataract_semantic_segmentation_amd import ops
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    with set_plan(plan, batch=int(orc["x"].shape[0])):
        model = OCRNet(dict(orc["cfg"]), 3)
        model.load_state_dict(fill_state(orc["spec"], orc["wseed"]))
        model.cuda().train()
        crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                             "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
        ops.PROFILE = []
        interm, final = model(orc["x"].cuda())
        loss = crit(interm, final, orc["lbl"].cuda())
        loss.backward()
        torch.cuda.synchronize()
        kinds = {k for k, *_ in ops.PROFILE}
        ops.PROFILE = None
    return model, interm.detach().cpu(), final.detach().cpu(), float(loss), kinds


def hrnet48_figures(orc, interm_h, final_h, loss_h):
    f32, f64 = orc["final32"], orc["final64"]
    fig = {
        "logit_scale": float(f32.abs().max()),
        "e_abs_vs_cpu32": float((final_h - f32).abs().max()),
        "e_abs_interm_vs_cpu32": float((interm_h - orc["interm32"]).abs().max()),
        "e_abs_vs_fp64": float((final_h.double() - f64).abs().max()),
        "e_abs_interm_vs_fp64": float((interm_h.double() - orc["interm64"]).abs().max()),
        "cpu32_e_abs_vs_fp64": float((f32.double() - f64).abs().max()),
        "cpu32_e_abs_interm_vs_fp64": float((orc["interm32"].double() - orc["interm64"]).abs().max()),
        "e_rms_vs_fp64": float((final_h.double() - f64).pow(2).mean().sqrt()),
        "cpu32_e_rms_vs_fp64": float((f32.double() - f64).pow(2).mean().sqrt()),
        "loss_hip": loss_h, "loss_cpu32": orc["loss32"], "loss_fp64": orc["loss64"],
    }
    fig["argmax"] = argmax_figures(final_h, f32, f64)
    return fig
