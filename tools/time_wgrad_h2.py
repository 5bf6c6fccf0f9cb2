"""backward-weight of the head layers (f16x2): launch time; CATSEG_LIB selects the library"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
out = "%-28s" % os.path.basename(os.environ.get("CATSEG_LIB", "default"))
for (B, H, W, Ci, Co, k, p) in [(8, 136, 240, 720, 512, 3, 1), (8, 136, 240, 1024, 512, 1, 0), (8, 68, 120, 2048, 512, 3, 1)]:
    x = torch.randn(B, H, W, Ci, device=dev); dy = torch.randn(B, H, W, Co, device=dev) * 1e-4
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    dw = torch.empty_like(w)
    ops.PROFILE = []
    for _ in range(8):
        ops.conv_bwd_weight(x, dy, dw, None, k, k, 1, p, 1)
    torch.cuda.synchronize()
    t = [e0.elapsed_time(e1) for kind, fl, e0, e1 in ops.PROFILE if kind == "wgrad_h2"][3:]
    ops.PROFILE = None
    out += "  %d->%d k%d: %.3f ms" % (Ci, Co, k, sum(t) / len(t))
print(out, flush=True)
