"""What would a MULTI-PROBLEM launch of the direct trunk kernels buy?  A launch of catseg_dconv3_pl / catseg_dwgrad3_pl is ONE wave of blocks
(~510 tiles on 512 block slots): its prologue (first LDS-DMA round trip), its epilogue and the launch itself are paid in full, nothing
overlaps them.  This times eight INDEPENDENT problems of one width (a) back to back on one stream (what tools/time_pl.py reports) and (b) spread
over 2 / 4 streams, where the hardware may start a problem's blocks while the previous problem's last blocks drain -- a lower bound of what one
launch over the eight problems' tiles would reach.   usage: python tools/time_multi_stream.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
streams = [torch.cuda.Stream(dev) for _ in range(4)]


def timed(fn, nstreams):
    """fn(i) for i in 0..7, problem i on stream i % nstreams; us per problem"""
    main = torch.cuda.current_stream(dev)

    def run():
        ev = torch.cuda.Event()
        ev.record(main)
        for s in streams[:nstreams]:
            s.wait_event(ev)
        for i in range(8):
            with torch.cuda.stream(streams[i % nstreams]):
                fn(i)
        for s in streams[:nstreams]:
            main.wait_stream(s)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (8 * reps) * 1e3


for (B, H, W, C) in [(8, 68, 120, 96), (8, 34, 60, 192), (8, 17, 30, 384)]:
    xs = [torch.randn(B, H, W, C, device=dev) for _ in range(8)]
    dys = [torch.randn(B, H, W, C, device=dev) * 1e-3 for _ in range(8)]
    for t in xs + dys:
        t._amax = ops.new_amax(dev)
        t._amax[0:1] = t.abs().max().reshape(1).view(torch.int32)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    wimg = ops.dconv3_weight_image(w, h2=True)
    xps = [ops.planes_from_f32(x, x._amax) for x in xs]
    dps = [ops.planes_from_f32(d, d._amax) for d in dys]
    ys = [torch.empty_like(xs[0]) for _ in range(8)]
    dws = [torch.empty_like(w) for _ in range(8)]
    # (workspaces are keyed by (device, stream): concurrent problems on different streams do not share slabs / partial buffers)
    fwd = [timed(lambda i: ops.dconv3_pl(xps[i], wimg, None, out=ys[i], bn_stats=True), n) for n in (1, 2, 4)]
    wg = [timed(lambda i: ops.dwgrad3_pl(xps[i], dps[i], dws[i]), n) for n in (1, 2, 4)]
    print("C=%3d %dx%dx%d   forward + BN partials: 1 stream %.1f us, 2 streams %.1f, 4 streams %.1f   |   backward-weight + reduction: %.1f / %.1f / %.1f"
          % (C, B, H, W, fwd[0], fwd[1], fwd[2], wg[0], wg[1], wg[2]), flush=True)
    ops.release_b3_cache()
