#!/bin/bash
# on the GPU box: time the direct 3x3 kernel with every ab/libcatseg_dc_*.so
R=${GRAFT_REPO_ROOT:-$PWD}
for f in $R/ab/libcatseg_dc_*.so; do
  CATSEG_LIB=$f python3 - "$f" <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd._lib import lib
dev = torch.device("cuda")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = "%-28s" % os.path.basename(sys.argv[1])
for (B, H, W, C) in [(8, 136, 240, 48), (8, 68, 120, 96), (8, 34, 60, 192), (8, 17, 30, 384)]:
    if not lib.catseg_dconv3_supported(C):
        continue
    x = torch.randn(B, H, W, C, device=dev)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x)
    wimg = ops.dconv3_weight_image(w)
    tf = timeit(lambda: ops.dconv3(x, wimg, None, out=y, bn_stats=True))
    td = timeit(lambda: ops.dconv3(x, wimg, None, out=y))
    gf = 2.0 * B * H * W * C * C * 9 / 1e9
    out += "  C=%d: fwd+bn %6.1f us %5.0f TF | plain %6.1f us %5.0f TF" % (C, tf, gf / tf * 1e3, td, gf / td * 1e3)
print(out, flush=True)
PY
done
