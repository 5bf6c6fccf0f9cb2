#!/bin/bash
# on the GPU box: bf16x3 backward-weight (igemm_b3t_kernel) of the large layers with every ab/libcatseg_t_*.so
R=${GRAFT_REPO_ROOT:-$PWD}
for f in $R/ab/libcatseg_t_*.so; do
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
for (B, H, W, Ci, Co, k, p, d) in [(8, 136, 240, 720, 512, 3, 1, 1), (8, 68, 120, 2048, 512, 3, 1, 1), (8, 68, 120, 512, 512, 3, 4, 4)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = torch.empty(Co, Ci, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, W, Co, device=dev)
    t = timeit(lambda: ops.conv_bwd_weight(x, dy, w, None, k, k, 1, p, d))       # planes cached after the first call
    fl = 2.0 * B * H * W * Co * Ci * k * k
    print("  %dx%d d%-2d %4d->%4d @%dx%d: wgrad %.3f ms %.0f TF-eq" % (k, k, d, Ci, Co, H, W, t, fl / t / 1e9), flush=True)
PY
done
