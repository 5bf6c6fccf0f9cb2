"""Full-resolution parity of the bench model (OCRNet-HRNet-W48, 2 x 3 x 544 x 960) under every arithmetic plan, one oracle evaluation:
writes gpurun_out/parity_fullres.json (copied to profiles/r04_parity_fullres.json).  Usage: python tools/parity_fullres.py [plan ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import _fullres as FR  # noqa: E402


def main():
    plans = sys.argv[1:] or list(FR.PLANS)
    t0 = time.time()
    orc = FR.hrnet48_oracle()
    print("oracle: %.0f s; cpu32 vs fp64 %.3g" % (time.time() - t0, float((orc["final32"].double() - orc["final64"]).abs().max())), flush=True)
    for plan in plans:
        model, interm, final, loss, kinds = FR.hrnet48_hip(orc, plan)
        fig = FR.hrnet48_figures(orc, interm, final, loss)
        fig["kernel_populations"] = sorted(k for k in kinds if not k.startswith("hbm:"))
        FR.record("ocrnet_hrnet48_2x544x960", plan, fig)
        print(plan, {k: (round(v, 7) if isinstance(v, float) else v) for k, v in fig.items()}, flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
