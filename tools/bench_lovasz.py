"""Lovasz-Softmax at the configuration's size (P = 8 x 544 x 960, K = 25): event time per loss call for random-init-like and
trained-like (confident) logits, active-set pruning on / off.  usage: bench_lovasz.py [n]   (run under rocprofv3 for the kernel split)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda")
B, H, W, K = 8, 544, 960, 25
P = B * H * W
g = torch.Generator(device="cuda").manual_seed(1)
lbl = torch.randint(0, K + 1, (B, H // 16, W // 16), device=dev, generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2).reshape(-1).contiguous()
noise = torch.randn(P, K, device=dev, generator=g)
onehot = torch.nn.functional.one_hot(lbl.clamp(max=K - 1), K).float()
cases = {"random-init-like (N(0,1) logits)": noise.clone(), "trained-like (logit margin 8 + N(0,1))": (8.0 * onehot + noise).contiguous()}
dl = torch.empty(P, K, device=dev)
for prune in (1, 0):
    _lib.lib.catseg_debug_set_lovasz_prune(prune)
    for name, lg in cases.items():
        ops.lovasz_softmax(lg, lbl, 1.0, dl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            loss = ops.lovasz_softmax(lg, lbl, 1.0, dl)
        e1.record(); torch.cuda.synchronize()
        print("prune %d  %-42s %.3f ms per loss call (loss %.6f)" % (prune, name, e0.elapsed_time(e1) / n, float(loss)), flush=True)
_lib.lib.catseg_debug_set_lovasz_prune(1)
