import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for xs, Cout, k, s, pd, name in [((8,136,240,64),256,1,1,0,"64->256"), ((8,136,240,256),64,1,1,0,"256->64"), ((8,136,240,64),64,3,1,1,"64->64 3x3"), ((8,272,480,64),64,3,2,1,"stem conv2"), ((8,136,240,256),48,3,1,1,"256->48 3x3")]:
    B,H,W,Cin = xs
    x = [torch.randn(B,H,W,Cin,device=dev).relu_() for _ in range(4)]
    w = (torch.randn(Cout,Cin,k,k,device=dev)*0.05).contiguous(memory_format=torch.channels_last)
    out = []
    for mi, ni in [(0,0),(1,1),(2,1),(1,2),(2,2),(4,1),(4,2),(2,4)]:
        ops.lib.catseg_debug_set_tile(mi, ni)
        try:
            t = timeit(lambda i: ops.conv_fwd(x[i%4], w, None, Cout, k, k, s, pd, 1, bn_stats=True, exact=True))
            out.append("(%d,%d) %.1f" % (mi, ni, t))
        except Exception as e:
            out.append("(%d,%d) err" % (mi, ni))
    ops.lib.catseg_debug_set_tile(0, 0)
    print(name, "fwd us:", "  ".join(out), flush=True)
