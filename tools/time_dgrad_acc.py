import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
B, H, W, Ci, Co, k, p = 8, 136, 240, 720, 512, 3, 1
x = torch.randn(B, H, W, Ci, device=dev)
w = (torch.randn(Co, Ci, k, k, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, H, W, Co, device=dev) * 1e-4
dx = torch.zeros_like(x)
for acc in (False, True, False, True):
    ops.PROFILE = []
    for _ in range(6):
        ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, 1, p, 1, out=dx, accumulate=acc)
    torch.cuda.synchronize()
    t = [e0.elapsed_time(e1) for kind, fl, e0, e1 in ops.PROFILE if kind == "dgrad_h2"][2:]
    ops.PROFILE = None
    print("accumulate=%s: %.3f ms" % (acc, sum(t) / len(t)), flush=True)
