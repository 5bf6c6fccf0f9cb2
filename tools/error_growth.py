"""Where does the forward error enter?  OCRNet-HRNet-W48 train-mode forward at 2 x 3 x H x W: relative RMS distance to the fp64 oracle of
named intermediate activations (engine.tap / oracle.nets.tap) for the fp32 CPU oracle and for the HIP path under each arithmetic plan.
Usage: python tools/error_growth.py [H W] [plan ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import _fullres as FR  # noqa: E402


def main():
    args = sys.argv[1:]
    H, W = (int(args[0]), int(args[1])) if len(args) >= 2 and args[0].isdigit() else (544, 960)
    plans = [a for a in args if not a.isdigit()] or ["production", "fp32", "all_bf16x3"]
    import bench
    from oracle import nets as ON
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd import engine
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    cfg = dict(bench.MODELS["ocrnet_hrnet48"][0])
    spec = spec_of(OCRNet(dict(cfg), 3).state_dict())
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, H, W, generator=g)
    taps = {}
    with torch.no_grad():
        ON.TAPS = taps["cpu32"] = {}
        ON.ocrnet_hrnet_forward(fill_state(spec, 41), x, train=True)
        ON.TAPS = taps["fp64"] = {}
        S64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, 41).items()}
        ON.ocrnet_hrnet_forward(S64, x.double(), train=True)
        ON.TAPS = None
        for plan in plans:
            with FR.set_plan(plan, batch=2):
                model = OCRNet(dict(cfg), 3)
                model.load_state_dict(fill_state(spec, 41))
                model.cuda().train()
                engine.TAPS = taps[plan] = {}
                model(x.cuda())
                torch.cuda.synchronize()
                engine.TAPS = None
                del model
    names = list(taps["fp64"])
    cols = ["cpu32"] + plans
    print("%-16s %10s " % ("tap", "rms(fp64)") + " ".join("%12s" % c for c in cols) + "   (relative RMS error vs fp64; ratio to cpu32 in brackets)")
    out = {}
    for n in names:
        ref = taps["fp64"][n]
        K = ref.shape[1]
        rms = float(ref.pow(2).mean().sqrt())
        errs = []
        for c in cols:
            t = taps[c][n][:, :K].double()
            errs.append(float((t - ref).pow(2).mean().sqrt()) / rms)
        out[n] = dict(zip(cols, errs))
        print("%-16s %10.3g " % (n, rms) + " ".join("%8.3g(%3.1f)" % (e, e / errs[0]) for e in errs))
    FR.record("error_growth_%dx%d" % (H, W), "relative_rms_vs_fp64", out)


if __name__ == "__main__":
    main()
