"""standalone times of the f16x2 direct kernels (forward + BN partials, backward-data) at the trunk shapes; CATSEG_LIB selects the library"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = "%-24s" % os.path.basename(os.environ.get("CATSEG_LIB", "default"))
for (B, H, W, C) in [(8, 136, 240, 48), (8, 68, 120, 96), (8, 34, 60, 192), (8, 17, 30, 384), (8, 136, 240, 64)]:
    # (eight different inputs in turn: a single tensor would sit in the 256 MB last-level cache, which the step's tensors do not)
    xs = [torch.randn(B, H, W, C, device=dev) for _ in range(8)]
    for x in xs:
        x._amax = ops.new_amax(dev); x._amax[0:1] = x.abs().max().reshape(1).view(torch.int32)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(xs[0])
    wimg, wimg_t = ops.dconv3_weight_image(w, h2=True), ops.dconv3_weight_image(w, backward_data=True, h2=True)
    it = [0]
    def fwd():
        x = xs[it[0] % 8]; it[0] += 1
        ops.dconv3(x, wimg, None, out=y, bn_stats=True, x_amax=x._amax)
    def bwd():
        x = xs[it[0] % 8]; it[0] += 1
        ops.dconv3(x, wimg_t, None, out=y, x_amax=x._amax)
    out += "  C=%d: %5.1f / %5.1f us" % (C, timeit(fwd), timeit(bwd))
print(out, flush=True)
