"""A/B of the head layers' forward / backward-data kernel (csrc/igemm_f16x2.hip): igemm_h2w8_kernel (two waves per SIMD, 8 waves per block) against
igemm_h2w_kernel (one 512-register wave per SIMD) -- standalone times of the three head layer shapes in both directions with bit-identity of
the results (outputs and BatchNorm partials), then the HRNet-W48 train step replayed as a hipGraph in alternating rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd._lib import lib
from miccai2021_cataract_semantic_segmentation_amd.graph import GraphedTrainStep
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
dev = torch.device("cuda")


def ev_time(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, H, W, Ci, Co, k, p) in [(8, 136, 240, 720, 512, 3, 1), (8, 136, 240, 1024, 512, 1, 0), (2, 37, 51, 720, 512, 3, 1)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, W, Co, device=dev) * 1e-4
    res = {}
    for waves in (4, 8, 4, 8):
        lib.catseg_debug_set_h2w_waves(waves)
        ops.release_b3_cache()
        y = torch.empty(B, H, W, Co, device=dev)
        dx = torch.empty_like(x)
        ops.PROFILE = []
        for _ in range(5):
            ops.release_b3_cache()
            r = ops.conv_fwd(x, w, None, Co, k, k, 1, p, 1, out=y, train=True, bn_stats=True)
            ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, 1, p, 1, out=dx)
        torch.cuda.synchronize()
        agg = {}
        for kind, fl, e0, e1 in ops.PROFILE[len(ops.PROFILE) // 5 * 2:]:
            a = agg.setdefault(kind, [0.0, 0, fl]); a[0] += e0.elapsed_time(e1); a[1] += 1
        ops.PROFILE = None
        part = r[1][0] if isinstance(r, tuple) else None
        cur = (y.clone(), dx.clone(), None if part is None else part.clone())
        if waves in res:
            pass
        res.setdefault(waves, cur)
        print("%dx%dx%d %d->%d k%d waves %d: " % (B, H, W, Ci, Co, k, waves) +
              "  ".join("%s %.3f ms (%.0f TFLOP/s-eq)" % (kk, v[0] / v[1], v[2] / (v[0] / v[1] * 1e-3) / 1e12) for kk, v in sorted(agg.items())
                        if kk in ("fwd_h2", "dgrad_h2")), flush=True)
    a, b = res[4], res[8]
    print("   bit-identical: y %s  dx %s  bn partials %s" % (torch.equal(a[0], b[0]), torch.equal(a[1], b[1]),
                                                              None if a[2] is None else torch.equal(a[2], b[2])), flush=True)

torch.manual_seed(0)
model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=0.0)
batches = [bench.synth_batch(8, 544, 960, 25, 1000 + 7919 * i, dev) for i in range(4)]
for rnd in range(3):
    for waves in (8, 4):
        lib.catseg_debug_set_h2w_waves(waves)
        step = GraphedTrainStep(model, lambda o, l: crit(o[0], o[1], l), opt, *batches[0])
        for i in range(3):
            step(*batches[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            loss = step(*batches[i % 4])
        torch.cuda.synchronize()
        print("round %d waves %d: %.2f ms/step (loss %.7f)" % (rnd, waves, (time.perf_counter() - t0) / 10 * 1e3, float(loss)), flush=True)
        step.release()
        del step
lib.catseg_debug_set_h2w_waves(8)
