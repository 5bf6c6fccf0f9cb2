#!/bin/bash
# on the GPU box: blocked-plane bf16x3 forward / backward-data of the large layers with every ab/libcatseg_k_*.so
R=${GRAFT_REPO_ROOT:-$PWD}
for f in $R/ab/libcatseg_k_*.so; do
  CATSEG_LIB=$f python3 - "$f" <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(os.path.basename(sys.argv[1]))
for (B, H, W, Ci, Co, k, p, d) in [(8, 136, 240, 720, 512, 3, 1, 1), (8, 68, 120, 2048, 512, 3, 1, 1), (8, 68, 120, 512, 512, 3, 4, 4), (8, 68, 120, 2048, 256, 3, 12, 12)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, H, W, Co, device=dev); dx = torch.empty_like(x)
    xb, wb = ops.split3_blocked(x)[0], ops.split3_weight_blocked(w)
    dyb, wtb = ops.split3_blocked(y.normal_())[0], ops.split3_weight_t_blocked(w)
    tf = timeit(lambda: ops.conv_fwd_b3_blocked(tuple(x.shape), xb, wb, None, Co, k, k, 1, p, d, out=y))
    td = timeit(lambda: ops.conv_bwd_data_b3_blocked(dyb, wtb, tuple(x.shape), Co, k, k, 1, p, d, out=dx))
    fl = 2.0 * B * H * W * Co * Ci * k * k
    print("  %dx%d d%-2d %4d->%4d @%dx%d: fwd %.3f ms %.0f TF-eq | dgrad %.3f ms %.0f TF-eq" % (k, k, d, Ci, Co, H, W, tf, fl / tf / 1e9, td, fl / td / 1e9), flush=True)
PY
done
