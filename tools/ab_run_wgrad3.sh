#!/bin/bash
# on the GPU box: time the direct backward-weight kernel with every ab/libcatseg_wg_*.so (three rounds, interleaved)
R=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2 3; do
for f in $R/ab/libcatseg_wg_*.so; do
  CATSEG_LIB=$f python3 - "$f" <<'PY'
import sys, os
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
out = "%-28s" % os.path.basename(sys.argv[1])
for (B, H, W, C) in [(8, 136, 240, 48), (8, 68, 120, 96), (8, 34, 60, 192), (8, 17, 30, 384)]:
    x = torch.randn(B, H, W, C, device=dev); dy = torch.randn(B, H, W, C, device=dev)
    dw = torch.empty(C, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    t = timeit(lambda: ops.dwgrad3(x, dy, dw))
    out += "  C=%d %6.1f us" % (C, t)
print(out, flush=True)
PY
done; done
