#!/bin/bash
# on the GPU box: forward fp32 convolution of several shapes with ab/libcatseg_f32_base.so and ab/libcatseg_f32_blocked.so
R=${GRAFT_REPO_ROOT:-$PWD}
for f in $R/ab/libcatseg_f32_base.so $R/ab/libcatseg_f32_blocked.so; do
  CATSEG_LIB=$f CATSEG_PRECISION=fp32 python3 - "$f" <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(os.path.basename(sys.argv[1]))
for (B, H, W, Ci, Co, k, p, d) in [(8, 68, 120, 1024, 256, 1, 0, 1), (8, 68, 120, 256, 1024, 1, 0, 1), (8, 68, 120, 2048, 512, 1, 0, 1), (8, 68, 120, 256, 256, 3, 2, 2),
                                   (8, 136, 240, 512, 256, 1, 0, 1), (8, 136, 240, 256, 512, 1, 0, 1), (8, 136, 240, 48, 48, 3, 1, 1), (8, 68, 120, 96, 96, 3, 1, 1),
                                   (8, 34, 60, 192, 192, 3, 1, 1), (8, 68, 120, 512, 512, 3, 4, 4)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, H, W, Co, device=dev)
    t = timeit(lambda: ops.conv_fwd(x, w, None, Co, k, k, 1, p, d, out=y))
    fl = 2.0 * B * H * W * Co * Ci * k * k
    print("  %dx%d %4d->%4d @%dx%d d%d: %8.1f us  %6.1f TF" % (k, k, Ci, Co, H, W, d, t, fl / t / 1e6), flush=True)
PY
done
