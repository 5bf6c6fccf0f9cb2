"""Generates the committed fixtures in tests/golden/ by running the REAL reference
(imported through ref_harness.py) in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures are data only (inputs, seeds, expected outputs); no reference source travels.
Weights of the big nets are not stored: they are regenerated from (key, shape, seed)
by ``oracle.state.fill_state``; the (key, shape) spec itself is stored so that the
checkpoint-key contract is pinned too.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness  # noqa: E402
from oracle.state import fill_state, spec_of  # noqa: E402

torch.set_num_threads(8)
R = ref_harness.load()


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %s  %.1f KB" % (name, os.path.getsize(path) / 1024))


def grad_summary(model):
    """per-parameter (L2 norm, sum) of .grad + full grads of the small tensors"""
    names, norms, sums, small = [], [], [], {}
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(k)
        norms.append(float(g.double().norm()))
        sums.append(float(g.double().sum()))
        if g.numel() <= 4096:
            small["g:" + k] = npy(g)
    return names, np.array(norms), np.array(sums), small


def whole_net(kind, experiment, seed, shape, steps=3):
    torch.manual_seed(seed)
    if kind == "ocrnet":
        model = R.models.OCRNet({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, experiment)
        loss_fn = R.losses.TwoScaleLoss({"experiment": experiment,
                                         "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    else:
        model = R.models.DeepLabv3Plus({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, experiment)
        ce = torch.nn.CrossEntropyLoss(ignore_index={2: 17, 3: 25}.get(experiment, -100))
        loss_fn = lambda out, lbl: ce(out, lbl)  # noqa: E731
    spec = spec_of(model.state_dict())
    model.load_state_dict(fill_state(spec, seed))
    K = model.num_classes
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(shape, generator=g)
    lbl = torch.randint(0, K + (1 if experiment in (2, 3) else 0), (shape[0],) + tuple(shape[2:]), generator=g)
    # blob structure: make a couple of classes absent and some regions constant
    lbl[:, : shape[2] // 3, : shape[3] // 2] = 0
    lbl[lbl == 3] = 4
    out = {"x": npy(x), "lbl": npy(lbl).astype(np.int64), "seed": np.array(seed),
           "spec": np.array(json.dumps(spec))}
    # eval-mode forward first (does not touch running stats)
    model.eval()
    with torch.no_grad():
        if kind == "ocrnet":
            i_e, o_e = model(x)
            out["eval_interm"], out["eval_final"] = npy(i_e), npy(o_e)
        else:
            out["eval_final"] = npy(model(x))
    # training trace: Adam lr 1e-4 (managers/BaseManager.py:441), `steps` steps
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    losses = []
    for s in range(steps):
        opt.zero_grad()
        if kind == "ocrnet":
            interm, final = model(x)
            loss = loss_fn(interm, final, lbl)
        else:
            final = model(x)
            loss = loss_fn(final, lbl)
        loss.backward()
        if s == 0:
            out["train_final"] = npy(final)
            if kind == "ocrnet":
                out["train_interm"] = npy(interm)
            names, norms, sums, small = grad_summary(model)
            out["grad_names"] = np.array(json.dumps(names))
            out["grad_norms"], out["grad_sums"] = norms, sums
            out.update(small)
            sd = model.state_dict()
            for k in ("backbone.bn1.running_mean", "backbone.bn1.running_var",
                      "backbone.layer4.2.bn3.running_mean", "backbone.layer4.2.bn3.running_var"):
                out["rs:" + k] = npy(sd[k])
        opt.step()
        losses.append(float(loss))
    out["losses"] = np.array(losses)
    sd = model.state_dict()
    out["final_param_sums"] = np.array([float(v.double().sum()) for k, v in sd.items() if v.dtype.is_floating_point])
    save("%s_r50_e%d_tiny" % (kind, experiment), **out)


def lovasz_cases():
    out = {}
    L = R.losses.LovaszSoftmax({"experiment": 3})
    torch.manual_seed(0)
    lg = torch.randn(2, 25, 32, 48, requires_grad=True)
    lb = torch.randint(0, 26, (2, 32, 48))
    loss = L(lg, lb)
    loss.backward()
    out.update(a_logits=npy(lg), a_labels=npy(lb), a_loss=np.array(float(loss)), a_grad=npy(lg.grad))
    # blob-structured labels with absent classes, K=17 (experiment 2), odd sizes
    L2 = R.losses.LovaszSoftmax({"experiment": 2})
    g = torch.Generator().manual_seed(5)
    lg = (3 * torch.randn(3, 17, 21, 37, generator=g)).requires_grad_()
    lb = torch.zeros(3, 21, 37, dtype=torch.int64)
    lb[:, 5:15, 3:20] = 4
    lb[0, 10:, 20:] = 9
    lb[1, :4, :] = 17
    lb[2, 8:12, 30:] = 16
    loss = L2(lg, lb)
    loss.backward()
    out.update(b_logits=npy(lg), b_labels=npy(lb), b_loss=np.array(float(loss)), b_grad=npy(lg.grad))
    # 1x2x2x2 KAT
    L1 = R.losses.LovaszSoftmax({"experiment": 1})
    lg = torch.tensor([[[[2., 0.], [0., -1.]], [[0., 1.], [3., 0.5]]]], requires_grad=True)
    lb = torch.tensor([[[0, 1], [0, 1]]])
    loss = L1(lg, lb)
    loss.backward()
    out.update(c_logits=npy(lg), c_labels=npy(lb), c_loss=np.array(float(loss)), c_grad=npy(lg.grad))
    from losses.LovaszSoftmax import lovasz_grad
    out["lovasz_grad_1010"] = npy(lovasz_grad(torch.tensor([1., 0., 1., 0.])))
    # TwoScale
    T = R.losses.TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                               "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    g = torch.Generator().manual_seed(7)
    a = torch.randn(2, 25, 16, 24, generator=g)
    b = torch.randn(2, 25, 16, 24, generator=g)
    lb = torch.randint(0, 26, (2, 16, 24), generator=g)
    out.update(t_interm=npy(a), t_final=npy(b), t_labels=npy(lb), t_loss=np.array(float(T(a, b, lb))))
    # CE with ignore
    ce = torch.nn.CrossEntropyLoss(ignore_index=17)
    lg = torch.randn(2, 17, 9, 13, generator=g).requires_grad_()
    lb = torch.randint(0, 18, (2, 9, 13), generator=g)
    loss = ce(lg, lb)
    loss.backward()
    out.update(ce_logits=npy(lg), ce_labels=npy(lb), ce_loss=np.array(float(loss)), ce_grad=npy(lg.grad))
    save("losses", **out)


def metric_cases():
    out = {}
    g = torch.Generator().manual_seed(11)
    for exp, K in ((1, 8), (2, 17), (3, 25)):
        lg = torch.randn(2, K, 24, 40, generator=g)
        lb = torch.randint(0, K + (1 if exp > 1 else 0), (2, 24, 40), generator=g)
        lg[:, 2] += 1.0
        cm = R.utils.t_get_confusion_matrix(lg, lb)
        ious = R.utils.t_get_mean_iou(cm, exp, categories=True, rare=True)
        pa, pac = R.utils.t_get_pixel_accuracy(cm)
        out.update({"e%d_logits" % exp: npy(lg), "e%d_labels" % exp: npy(lb), "e%d_cm" % exp: npy(cm),
                    "e%d_miou" % exp: np.array([float(v) for v in ious]),
                    "e%d_pa" % exp: np.array([float(pa), float(pac)])})
    f = R.utils.LRFcts({"epochs": 50, "learning_rate": 1e-4, "lr_fct": "exponential", "lr_params": None,
                        "lr_restarts": [], "lr_restart_vals": 1, "lr_batchwise": False}, [], 50)
    out["lr_mult"] = np.array([f(e) for e in range(50)])
    save("metrics", **out)


def ocr_modules():
    """Reference OCR head modules with small channel counts: full params/inputs/outputs/grads."""
    from models.OCR import SpatialGatherModule, ObjectAttentionBlock2D, SpatialOCR_Module
    out = {}
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(2, 32, 6, 10, generator=g, requires_grad=True)
    logits = (2 * torch.randn(2, 5, 6, 10, generator=g)).requires_grad_()
    ctx = SpatialGatherModule(5)(feats, logits)
    w = torch.randn(ctx.shape, generator=g)
    (ctx * w).sum().backward()
    out.update(sg_feats=npy(feats), sg_logits=npy(logits), sg_out=npy(ctx), sg_w=npy(w),
               sg_dfeats=npy(feats.grad), sg_dlogits=npy(logits.grad))
    # SURVEY KAT
    f2 = torch.arange(8.).view(1, 2, 2, 2)
    p2 = torch.tensor([[[[0., 0.], [0., 0.]], [[10., 0.], [0., 0.]]]])
    out["sg_kat"] = npy(SpatialGatherModule(2)(f2, p2))
    torch.manual_seed(4)
    mod = SpatialOCR_Module(in_channels=32, key_channels=16, out_channels=32, scale=1, dropout=0.0)
    spec = spec_of(mod.state_dict())
    mod.load_state_dict(fill_state(spec, 4))
    mod.train()
    x = torch.randn(2, 32, 6, 10, generator=g, requires_grad=True)
    proxy = torch.randn(2, 32, 5, 1, generator=g, requires_grad=True)
    y = mod(x, proxy)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    out.update(ocr_spec=np.array(json.dumps(spec)), ocr_x=npy(x), ocr_proxy=npy(proxy), ocr_y=npy(y), ocr_w=npy(w),
               ocr_dx=npy(x.grad), ocr_dproxy=npy(proxy.grad))
    for k, p in mod.named_parameters():
        out["ocr_g:" + k] = npy(p.grad)
    save("ocr_modules", **out)


if __name__ == "__main__":
    lovasz_cases()
    metric_cases()
    ocr_modules()
    whole_net("ocrnet", 3, 100, (2, 3, 64, 96))
    whole_net("deeplab", 2, 200, (2, 3, 64, 96))
