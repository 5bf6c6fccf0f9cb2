"""times the LovaszSoftmax loss + gradient call (and the cross-entropy / confusion-matrix kernels sharing its staging code) at
the configuration's size; CATSEG_LIB selects the library (tools/ab_old_new.sh lovasz builds the previous one)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator(device="cuda").manual_seed(5)
out = "%-30s" % os.path.basename(os.environ.get("CATSEG_LIB", "default"))
for K in (25, 17):
    P = 8 * 544 * 960
    lb = torch.randint(0, K, (8, 17, 30), device=dev, generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2).reshape(-1)
    # a half-trained network: the labelled class leads by ~4 on most pixels (the pruning of the sort then does what it does in training)
    lg = 1.5 * torch.randn(P, K, device=dev, generator=g) + 4.0 * torch.nn.functional.one_hot(lb, K).float()
    dl = torch.empty_like(lg)
    t = timeit(lambda: ops.lovasz_softmax(lg, lb, 1.0, dl))
    loss = float(ops.lovasz_softmax(lg, lb, 1.0, dl))
    out += "  K=%d: lovasz %7.1f us (loss %.7f, |dl| %.6e)" % (K, t, loss, float(dl.double().abs().sum()))
    if K == 25:
        # a randomly initialised network (the bench's regime): near-uniform predictions, the pruning keeps ~1/K of the elements
        lg = 0.05 * torch.randn(P, K, device=dev, generator=g)
        t = timeit(lambda: ops.lovasz_softmax(lg, lb, 1.0, dl))
        loss = float(ops.lovasz_softmax(lg, lb, 1.0, dl))
        out += "  K=25 random-init: %7.1f us (loss %.7f, |dl| %.6e)" % (t, loss, float(dl.double().abs().sum()))
print(out, flush=True)
